// MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950.
//   C[M,N] = act(alpha*(A.B^T + A2.B2^T) + bias + rowbias) + residual       (fp16 in, fp32 acc)
// Tile BMxBNx32, 256 threads = 4 waves (2x2), each wave (BM/2)x(BN/2) of v_mfma_f32_16x16x32_f16.
// Operands are staged global -> VGPR -> LDS (16 B per lane, rows padded to 80 B so the
// ds_read_b128 fragment reads are at most 2-way conflicted), double-buffered in LDS with the next
// k-tile's global loads in flight under the current tile's MFMAs.  The MFMA is issued with the
// operands swapped (D^T = B.A^T) so each lane ends up holding 4 consecutive N for one M row:
// the epilogue then does 8-byte stores along the contiguous dimension.
#include "common.h"

#define LDSK 40  // 32 halfs of K + 8 pad (80-byte rows)

struct ConvRow {
    int b, oy, ox;
    bool valid;
};

template <int BM, int BN, bool CONV>
__global__ __launch_bounds__(256) void gemm_kernel(fd_gemm_desc p, int ntm, int ntn) {
    constexpr int TM = BM / 32, TN = BN / 32;     // 16x16 tiles per wave in M / N
    constexpr int AI = BM / 64, BI = BN / 64;     // 16-byte chunks per thread per k-tile
    __shared__ __attribute__((aligned(16))) f16 smem[2 * (BM + BN) * LDSK];
    f16* As = smem;
    f16* Bs = smem + 2 * BM * LDSK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;

    const int tile = xcd_remap(blockIdx.x, ntm * ntn);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int z = blockIdx.y;

    const f16* A = (const f16*)p.A + (int64_t)z * p.sA;
    const f16* B = (const f16*)p.B + (int64_t)z * p.sB;
    const f16* A2 = (const f16*)p.A2;
    const f16* B2 = (const f16*)p.B2;

    const int nk1 = (p.K + 31) >> 5;
    const int nk2 = (p.K2 + 31) >> 5;
    const int nk = nk1 + nk2;

    // per-thread chunk coordinates
    int arow[AI], brow[BI];
    const int kc = tid & 3;
    ConvRow crow[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        arow[i] = (tid >> 2) + i * 64;
        if (CONV) {
            const int m = m0 + arow[i];
            const int hw = p.Ho * p.Wo;
            crow[i].valid = m < p.M;
            const int mm = crow[i].valid ? m : 0;
            crow[i].b = mm / hw;
            const int r = mm - crow[i].b * hw;
            crow[i].oy = r / p.Wo;
            crow[i].ox = r - crow[i].oy * p.Wo;
        }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) brow[i] = (tid >> 2) + i * 64;

    const int cpt = CONV ? (p.Cin >> 5) : 1;  // k-tiles per filter tap

    f16x8 ra[AI], rb[BI];
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_tiles = [&](int kt) {
        if (CONV) {
            const int tap = kt / cpt;
            const int c0 = (kt - tap * cpt) << 5;
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                int iy = crow[i].oy, ix = crow[i].ox;
                bool ok = crow[i].valid;
                if (p.conv_mode == FD_CONV_NORMAL) {
                    iy += ky - 1; ix += kx - 1;
                    ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                } else if (p.conv_mode == FD_CONV_STRIDE2) {
                    iy = 2 * iy + ky - 1; ix = 2 * ix + kx - 1;
                    ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                } else if (p.conv_mode == FD_CONV_UP2) {
                    iy += ky - 1; ix += kx - 1;
                    ok = ok && iy >= 0 && iy < 2 * p.H && ix >= 0 && ix < 2 * p.W;
                    iy >>= 1; ix >>= 1;
                } else {  // FD_CONV_TRANS2: data-gradient of the stride-2 conv (weights pre-flipped)
                    iy += ky - 1; ix += kx - 1;
                    ok = ok && iy >= 0 && ix >= 0 && !(iy & 1) && !(ix & 1);
                    iy >>= 1; ix >>= 1;
                    ok = ok && iy < p.H && ix < p.W;
                }
                ra[i] = zero8;
                if (ok) ra[i] = *(const f16x8*)(A + (((int64_t)crow[i].b * p.H + iy) * p.W + ix) * p.lda + c0 + kc * 8);
            }
            const int kk = kt * 32 + kc * 8;
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const int n = n0 + brow[i];
                rb[i] = zero8;
                if (n < p.N) rb[i] = *(const f16x8*)(B + (int64_t)n * p.ldb + kk);
            }
        } else {
            const bool seg2 = kt >= nk1;
            const f16* Ap = seg2 ? A2 : A;
            const f16* Bp = seg2 ? B2 : B;
            const int64_t la = seg2 ? p.lda2 : p.lda, lb = seg2 ? p.ldb2 : p.ldb;
            const int Kseg = seg2 ? p.K2 : p.K;
            const int kk = (seg2 ? kt - nk1 : kt) * 32 + kc * 8;
            const bool kok = kk < Kseg;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int m = m0 + arow[i];
                ra[i] = zero8;
                if (kok && m < p.M) ra[i] = *(const f16x8*)(Ap + (int64_t)m * la + kk);
            }
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const int n = n0 + brow[i];
                rb[i] = zero8;
                if (kok && n < p.N) rb[i] = *(const f16x8*)(Bp + (int64_t)n * lb + kk);
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AI; ++i) *(f16x8*)(As + (buf * BM + arow[i]) * LDSK + kc * 8) = ra[i];
#pragma unroll
        for (int i = 0; i < BI; ++i) *(f16x8*)(Bs + (buf * BN + brow[i]) * LDSK + kc * 8) = rb[i];
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        f16x8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *(const f16x8*)(As + (buf * BM + wm * (BM / 2) + i * 16 + l15) * LDSK + lg * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *(const f16x8*)(Bs + (buf * BN + wn * (BN / 2) + j * 16 + l15) * LDSK + lg * 8);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m][n..n+3], m = m0+wm*BM/2+i*16+l15, n = n0+wn*BN/2+j*16+lg*4
    const f16* R = p.residual ? (const f16*)p.residual + (int64_t)z * p.sR : nullptr;
    const f16* RB = (const f16*)p.rowbias;
    const bool vec_ok = ((p.N & 3) == 0) && ((p.ldc & 3) == 0) && (!R || (p.ldr & 3) == 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / 2) + i * 16 + l15;
        if (m >= p.M) continue;
        const int rbrow = RB ? m / p.rows_per_batch : 0;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 16 + lg * 4;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[i][j][r] * p.alpha;
                if (n + r < p.N) {
                    if (p.bias) x += p.bias[n + r];
                    if (RB) x += (float)RB[(int64_t)rbrow * p.ld_rowbias + n + r];
                    x = apply_act(x, p.act);
                    if (R) x += (float)R[(int64_t)m * p.ldr + n + r];
                }
                v[r] = x;
            }
            if (p.out_dtype == FD_OUT_F32) {
                float* C = (float*)p.C + (int64_t)z * p.sC + (int64_t)m * p.ldc + n;
                if (vec_ok) *(f32x4*)C = (f32x4){v[0], v[1], v[2], v[3]};
                else
                    for (int r = 0; r < 4 && n + r < p.N; ++r) C[r] = v[r];
            } else {
                f16* C = (f16*)p.C + (int64_t)z * p.sC + (int64_t)m * p.ldc + n;
                if (vec_ok) *(f16x4*)C = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                else
                    for (int r = 0; r < 4 && n + r < p.N; ++r) C[r] = (f16)v[r];
            }
        }
    }
}

template <int BM, int BN>
static int launch(const fd_gemm_desc& d, hipStream_t s) {
    const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
    dim3 grid(ntm * ntn, d.batch > 0 ? d.batch : 1);
    if (d.conv) hipLaunchKernelGGL((gemm_kernel<BM, BN, true>), grid, dim3(256), 0, s, d, ntm, ntn);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, false>), grid, dim3(256), 0, s, d, ntm, ntn);
    return fd_check_launch("fd_gemm");
}

extern "C" int fd_gemm(const fd_gemm_desc* dp, void* stream) {
    fd_gemm_desc d = *dp;
    FD_REQUIRE(d.A && d.B && d.C, "fd_gemm: null operand");
    FD_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "fd_gemm: bad shape M=%d N=%d K=%d", d.M, d.N, d.K);
    FD_REQUIRE((d.K & 7) == 0 && (d.lda & 7) == 0 && (d.ldb & 7) == 0, "fd_gemm: K, lda, ldb must be multiples of 8");
    if (d.K2 > 0) {
        FD_REQUIRE(!d.conv, "fd_gemm: second K-slab not supported with conv gather");
        FD_REQUIRE(d.A2 && d.B2 && (d.K2 & 7) == 0 && (d.lda2 & 7) == 0 && (d.ldb2 & 7) == 0, "fd_gemm: bad second slab");
        FD_REQUIRE(d.batch <= 1, "fd_gemm: second slab is not batched");
    } else d.K2 = 0;
    if (d.conv) {
        FD_REQUIRE((d.Cin & 31) == 0 && d.K == 9 * d.Cin, "fd_gemm(conv): Cin must be a multiple of 32 and K == 9*Cin");
        FD_REQUIRE(d.M == d.Bn * d.Ho * d.Wo, "fd_gemm(conv): M != B*Ho*Wo");
        FD_REQUIRE(d.batch <= 1, "fd_gemm(conv): not batched");
    }
    if (d.rowbias) FD_REQUIRE(d.rows_per_batch > 0, "fd_gemm: rows_per_batch");
    hipStream_t s = (hipStream_t)stream;
    // tile choice: big tiles when the grid still fills 256 CUs a few times over
    const long t128 = (long)((d.M + 127) / 128) * ((d.N + 127) / 128) * (d.batch > 0 ? d.batch : 1);
    const bool n64 = (d.N % 128) != 0 && (d.N % 128) <= 64;  // e.g. N=320: 128x64 tiles waste nothing
    if (t128 >= 512 && !n64) return launch<128, 128>(d, s);
    const long t12864 = (long)((d.M + 127) / 128) * ((d.N + 63) / 64) * (d.batch > 0 ? d.batch : 1);
    if (t12864 >= 512) return launch<128, 64>(d, s);
    return launch<64, 64>(d, s);
}
