// LoRA weight gradients: skinny "TN" reductions over the token dimension.
//   G[n, r] += scale * sum_m X[m, n] * T[m, r]        (X,T fp16; G fp32 with arbitrary strides)
// dUp[N,r]  = dY^T . t   (X = dY, T = t = x.down^T)        dDown[r,K] = u^T . x  (X = x, T = u = dY.up)
// HBM-bound (reads X once); VALU FMAs, coalesced over n, T rows broadcast.  Two-pass with a fixed
// reduction order so repeated runs are bit-identical.
#include "common.h"

template <int RP>
__global__ __launch_bounds__(256) void lora_wgrad_partial(const f16* __restrict__ X, int64_t ldx, const f16* __restrict__ T, int64_t ldt,
                                                          float* __restrict__ partial, int M, int N, int rows_per_split) {
    __shared__ float red[3][64][RP + 1];
    const int nl = threadIdx.x & 63, mg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + nl;
    const int m0 = blockIdx.y * rows_per_split;
    const int m1 = min(m0 + rows_per_split, M);
    float acc[RP];
#pragma unroll
    for (int r = 0; r < RP; ++r) acc[r] = 0.f;
    if (n < N) {
        for (int m = m0 + mg; m < m1; m += 4) {
            const float x = (float)X[(int64_t)m * ldx + n];
            const f16* tr = T + (int64_t)m * ldt;
#pragma unroll
            for (int r8 = 0; r8 < RP; r8 += 8) {
                const f16x8 tv = *(const f16x8*)(tr + r8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[r8 + j] += x * (float)tv[j];
            }
        }
    }
    if (mg > 0) {
#pragma unroll
        for (int r = 0; r < RP; ++r) red[mg - 1][nl][r] = acc[r];
    }
    __syncthreads();
    if (mg == 0 && n < N) {
#pragma unroll
        for (int r = 0; r < RP; ++r) {
            const float v = acc[r] + red[0][nl][r] + red[1][nl][r] + red[2][nl][r];
            partial[((int64_t)blockIdx.y * N + n) * RP + r] = v;
        }
    }
}

// Vectorised variant for small ranks: a thread owns CV consecutive columns (16-byte loads of X for CV = 8) and one row phase; a block
// covers CGB column groups x RT = 256 / CGB row phases (CGB = 40: 320 columns, the U-Net's channel granule).  The RT partial sums are
// combined through LDS in a fixed order.  Reads X at full line width instead of 2 bytes per lane.
template <int RP, int CV, int CGB>
__device__ __forceinline__ void lora_wgrad_partial_vec_body(const f16* __restrict__ X, int64_t ldx, const f16* __restrict__ T, int64_t ldt,
                                                            float* __restrict__ partial, int M, int N, int rows_per_split, int bx, int by,
                                                            float* red) {
    constexpr int RT = 256 / CGB;
    const int cgl = threadIdx.x % CGB, rt = threadIdx.x / CGB;
    const int n0 = (bx * CGB + cgl) * CV;
    const int m0 = by * rows_per_split;
    const int m1 = min(m0 + rows_per_split, M);
    float acc[CV][RP];
#pragma unroll
    for (int c = 0; c < CV; ++c)
#pragma unroll
        for (int r = 0; r < RP; ++r) acc[c][r] = 0.f;
    const bool active = rt < RT && n0 < N;      // N % CV == 0 is required by the launcher
    if (active) {
        constexpr int U = 4;                       // rows in flight per thread: the loop is latency-bound without this
        for (int m = m0 + rt; m < m1; m += U * RT) {
            f16 xv[U][CV];
            f16x8 tv[U][RP / 8];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int mm = m + u * RT;
                const bool ok = mm < m1;
                const int64_t mc = ok ? mm : m;   // clamp: the duplicate row is multiplied by zero below
                if (CV == 8) *(f16x8*)xv[u] = *(const f16x8*)(X + mc * ldx + n0);
                else if (CV == 4) *(f16x4*)xv[u] = *(const f16x4*)(X + mc * ldx + n0);
                else
#pragma unroll
                    for (int c = 0; c < CV; ++c) xv[u][c] = X[mc * ldx + n0 + c];
#pragma unroll
                for (int r8 = 0; r8 < RP / 8; ++r8) tv[u][r8] = *(const f16x8*)(T + mc * ldt + r8 * 8);
                if (!ok)
#pragma unroll
                    for (int c = 0; c < CV; ++c) xv[u][c] = (f16)0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float tf[RP];
#pragma unroll
                for (int r = 0; r < RP; ++r) tf[r] = (float)tv[u][r >> 3][r & 7];
#pragma unroll
                for (int c = 0; c < CV; ++c) {
                    const float x = (float)xv[u][c];
#pragma unroll
                    for (int r = 0; r < RP; ++r) acc[c][r] += x * tf[r];
                }
            }
        }
    }
    if (rt > 0 && rt < RT) {
        float* dst = red + ((rt - 1) * CGB + cgl) * (CV * RP + 1);
#pragma unroll
        for (int c = 0; c < CV; ++c)
#pragma unroll
            for (int r = 0; r < RP; ++r) dst[c * RP + r] = acc[c][r];
    }
    __syncthreads();
    if (rt == 0 && n0 < N) {
#pragma unroll
        for (int c = 0; c < CV; ++c)
#pragma unroll
            for (int r = 0; r < RP; ++r) {
                float v = acc[c][r];
#pragma unroll
                for (int k = 0; k < RT - 1; ++k) v += red[(k * CGB + cgl) * (CV * RP + 1) + c * RP + r];
                partial[((int64_t)by * N + n0 + c) * RP + r] = v;
            }
    }
}

template <int RP, int CV, int CGB>
__global__ __launch_bounds__(256) void lora_wgrad_partial_vec(const f16* __restrict__ X, int64_t ldx, const f16* __restrict__ T, int64_t ldt,
                                                              float* __restrict__ partial, int M, int N, int rows_per_split) {
    __shared__ float red[(256 / CGB - 1) * CGB * (CV * RP + 1)];
    lora_wgrad_partial_vec_body<RP, CV, CGB>(X, ldx, T, ldt, partial, M, N, rows_per_split, blockIdx.x, blockIdx.y, red);
}

// ---- batched form: up to FD_WGRAD_MAX independent problems (the 16 LoRA weight gradients of one transformer block's backward) in ONE partial
// launch and ONE final launch.  7808 + 7808 launches of ~10 + 7 us per training step were 3.7 % of it; the problems are far too small to fill the
// chip one at a time.  Same arithmetic and the same fixed reduction order per problem as the single-problem kernels.
struct WgradBatch {
    const f16* X[FD_WGRAD_MAX];
    const f16* T[FD_WGRAD_MAX];
    float* G[FD_WGRAD_MAX];
    int64_t ldx[FD_WGRAD_MAX], ldt[FD_WGRAD_MAX], sn[FD_WGRAD_MAX], sr[FD_WGRAD_MAX], poff[FD_WGRAD_MAX];
    int M[FD_WGRAD_MAX], N[FD_WGRAD_MAX], R[FD_WGRAD_MAX], rows[FD_WGRAD_MAX], ncb[FD_WGRAD_MAX], nsplit[FD_WGRAD_MAX];
    float scale[FD_WGRAD_MAX];
    int bstart[FD_WGRAD_MAX + 1];      // first block of each problem in the partial launch
    int ostart[FD_WGRAD_MAX + 1];      // first output element (n, r) of each problem in the final launch
    int n;
};

template <int RP, int CV, int CGB>
__global__ __launch_bounds__(256) void lora_wgrad_partial_multi(WgradBatch b, float* __restrict__ scratch) {
    FD_WG_TRACE(18);
    __shared__ float red[(256 / CGB - 1) * CGB * (CV * RP + 1)];
    int p = 0;
#pragma unroll 1
    while (p + 1 < b.n && (int)blockIdx.x >= b.bstart[p + 1]) ++p;
    const int local = blockIdx.x - b.bstart[p];
    const int bx = local % b.ncb[p], by = local / b.ncb[p];
    lora_wgrad_partial_vec_body<RP, CV, CGB>(b.X[p], b.ldx[p], b.T[p], b.ldt[p], scratch + b.poff[p], b.M[p], b.N[p], b.rows[p], bx, by, red);
}

__global__ __launch_bounds__(256) void lora_wgrad_final_multi(WgradBatch b, const float* __restrict__ scratch, int RP) {
    FD_WG_TRACE(19);
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= b.ostart[b.n]) return;
    int p = 0;
#pragma unroll 1
    while (p + 1 < b.n && o >= b.ostart[p + 1]) ++p;
    const int lo = o - b.ostart[p];
    const int lane = threadIdx.x & 63;
    const int R = b.R[p], N = b.N[p];
    const int n = lo / R, r = lo % R;
    const float* partial = scratch + b.poff[p];
    float s = 0.f;
    for (int k = lane; k < b.nsplit[p]; k += 64) s += partial[((int64_t)k * N + n) * RP + r];
    s = wave_sum(s);
    if (lane == 0) b.G[p][n * b.sn[p] + r * b.sr[p]] += b.scale[p] * s;
}

// one wave per output element (n, r): lanes stride over the splits in a fixed order, then a fixed-shape wave reduction
__global__ __launch_bounds__(256) void lora_wgrad_final(const float* partial, float* G, int64_t sn, int64_t sr, int N, int R, int RP, int nsplit,
                                                        float scale) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= N * R) return;
    const int lane = threadIdx.x & 63;
    const int n = o / R, r = o % R;
    float s = 0.f;
    for (int k = lane; k < nsplit; k += 64) s += partial[((int64_t)k * N + n) * RP + r];
    s = wave_sum(s);
    if (lane == 0) G[n * sn + r * sr] += scale * s;
}

extern "C" int fd_lora_wgrad(const void* X, int64_t ldx, const void* T, int64_t ldt, float* G, int64_t g_stride_n, int64_t g_stride_r, int M,
                             int N, int R, float scale, float* scratch, int64_t scratch_elems, void* stream) {
    FD_REQUIRE(M > 0 && N > 0 && R > 0 && R <= 64, "fd_lora_wgrad: rank must be in 1..64 (got %d)", R);
    const int RP = R <= 8 ? 8 : (R <= 16 ? 16 : (R <= 32 ? 32 : 64));
    FD_REQUIRE(ldt >= RP && (ldt & 7) == 0, "fd_lora_wgrad: T must be padded to %d columns (ldt=%ld)", RP, (long)ldt);
    // small ranks: vectorised kernel, 320 columns per block; otherwise 64 columns per block
    const bool vec = (RP == 8 && (N & 7) == 0 && (ldx & 7) == 0) || (RP == 16 && (N & 3) == 0 && (ldx & 3) == 0);
    const int cols_per_block = vec ? (RP == 8 ? 320 : 160) : 64;
    const int ncb = (N + cols_per_block - 1) / cols_per_block;
    int nsplit = (768 + ncb - 1) / ncb;
    if (nsplit > (M + 63) / 64) nsplit = (M + 63) / 64;
    while (nsplit > 1 && (int64_t)nsplit * N * RP > scratch_elems) nsplit >>= 1;
    FD_REQUIRE((int64_t)nsplit * N * RP <= scratch_elems, "fd_lora_wgrad: scratch too small");
    int rows = (M + nsplit - 1) / nsplit;
    rows = vec ? (rows + 5) / 6 * 6 : (rows + 3) & ~3;
    nsplit = (M + rows - 1) / rows;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(ncb, nsplit);
    switch (vec ? -RP : RP) {
        case -8: hipLaunchKernelGGL((lora_wgrad_partial_vec<8, 8, 40>), grid, dim3(256), 0, s, (const f16*)X, ldx, (const f16*)T, ldt, scratch, M, N, rows); break;
        case -16: hipLaunchKernelGGL((lora_wgrad_partial_vec<16, 4, 40>), grid, dim3(256), 0, s, (const f16*)X, ldx, (const f16*)T, ldt, scratch, M, N, rows); break;
        case 8: hipLaunchKernelGGL(lora_wgrad_partial<8>, grid, dim3(256), 0, s, (const f16*)X, ldx, (const f16*)T, ldt, scratch, M, N, rows); break;
        case 16: hipLaunchKernelGGL(lora_wgrad_partial<16>, grid, dim3(256), 0, s, (const f16*)X, ldx, (const f16*)T, ldt, scratch, M, N, rows); break;
        case 32: hipLaunchKernelGGL(lora_wgrad_partial<32>, grid, dim3(256), 0, s, (const f16*)X, ldx, (const f16*)T, ldt, scratch, M, N, rows); break;
        default: hipLaunchKernelGGL(lora_wgrad_partial<64>, grid, dim3(256), 0, s, (const f16*)X, ldx, (const f16*)T, ldt, scratch, M, N, rows); break;
    }
    hipLaunchKernelGGL(lora_wgrad_final, dim3((N * R + 3) / 4), dim3(256), 0, s, scratch, G, g_stride_n, g_stride_r, N, R, RP, nsplit, scale);
    return fd_check_launch("fd_lora_wgrad");
}

// Batched weight gradients: ``descs`` is a HOST array of n <= FD_WGRAD_MAX problems that share the padded rank RP (8 or 16) and satisfy the
// vectorised kernel's alignment (N %% CV == 0, ldx %% CV == 0 with CV = 8 / 4); anything else is rejected (callers fall back to fd_lora_wgrad).
extern "C" int fd_lora_wgrad_multi(const fd_wgrad_desc* descs, int n, float* scratch, int64_t scratch_elems, void* stream) {
    FD_REQUIRE(n >= 1 && n <= FD_WGRAD_MAX, "fd_lora_wgrad_multi: 1..%d problems (got %d)", FD_WGRAD_MAX, n);
    FD_REQUIRE_DESC(descs, fd_wgrad_desc, "fd_lora_wgrad_multi");       // element 0 first: its size is the array stride of the others
    for (int i = 1; i < n; ++i) FD_REQUIRE_DESC(descs + i, fd_wgrad_desc, "fd_lora_wgrad_multi");
    WgradBatch b;
    b.n = n;
    int RP = 0;
    double work = 0;
    for (int i = 0; i < n; ++i) {
        const fd_wgrad_desc& d = descs[i];
        FD_REQUIRE(d.M > 0 && d.N > 0 && d.R > 0 && d.R <= 16, "fd_lora_wgrad_multi: rank must be in 1..16");
        const int rp = d.R <= 8 ? 8 : 16;
        FD_REQUIRE(RP == 0 || rp == RP, "fd_lora_wgrad_multi: mixed padded ranks");
        RP = rp;
        const int cv = RP == 8 ? 8 : 4;
        FD_REQUIRE((d.N % cv) == 0 && (d.ldx % cv) == 0 && d.ldt >= RP && (d.ldt & 7) == 0, "fd_lora_wgrad_multi: alignment (N, ldx %% %d; ldt)", cv);
        work += (double)d.M * d.N;
    }
    const int cols_per_block = RP == 8 ? 320 : 160;
    int64_t poff = 0;
    int blocks = 0, outs = 0;
    for (int i = 0; i < n; ++i) {
        const fd_wgrad_desc& d = descs[i];
        const int ncb = (d.N + cols_per_block - 1) / cols_per_block;
        // ~1536 blocks over the whole batch, shared out by work; at least 64 rows per split
        int nsplit = (int)(1536.0 * ((double)d.M * d.N / work) / ncb + 0.5);
        if (nsplit < 1) nsplit = 1;
        if (nsplit > (d.M + 63) / 64) nsplit = (d.M + 63) / 64;
        int rows = (d.M + nsplit - 1) / nsplit;
        rows = (rows + 5) / 6 * 6;
        nsplit = (d.M + rows - 1) / rows;
        b.X[i] = (const f16*)d.X; b.T[i] = (const f16*)d.T; b.G[i] = d.G;
        b.ldx[i] = d.ldx; b.ldt[i] = d.ldt; b.sn[i] = d.g_stride_n; b.sr[i] = d.g_stride_r; b.poff[i] = poff;
        b.M[i] = d.M; b.N[i] = d.N; b.R[i] = d.R; b.rows[i] = rows; b.ncb[i] = ncb; b.nsplit[i] = nsplit; b.scale[i] = d.scale;
        b.bstart[i] = blocks; b.ostart[i] = outs;
        blocks += ncb * nsplit;
        outs += d.N * d.R;
        poff += (int64_t)nsplit * d.N * RP;
    }
    b.bstart[n] = blocks; b.ostart[n] = outs;
    FD_REQUIRE(poff <= scratch_elems, "fd_lora_wgrad_multi: scratch too small (%ld > %ld floats)", (long)poff, (long)scratch_elems);
    hipStream_t s = (hipStream_t)stream;
    if (RP == 8) hipLaunchKernelGGL((lora_wgrad_partial_multi<8, 8, 40>), dim3(blocks), dim3(256), 0, s, b, scratch);
    else hipLaunchKernelGGL((lora_wgrad_partial_multi<16, 4, 40>), dim3(blocks), dim3(256), 0, s, b, scratch);
    hipLaunchKernelGGL(lora_wgrad_final_multi, dim3((outs + 3) / 4), dim3(256), 0, s, b, (const float*)scratch, RP);
    return fd_check_launch("fd_lora_wgrad_multi");
}


// ------------------------------------------------------------------ 16-bit operand copies of LoRA pairs after an optimiser step
// 128 pairs x (zeros, two slice copies, two transposes) were ~1000 tiny torch launches per step (10 ms of host-bound time); here 16 pairs per launch.
#define FD_REFRESH_MAX 16
struct RefreshBatch { fd_lora_refresh_desc d[FD_REFRESH_MAX]; int n; };

__global__ void lora_refresh_kernel(RefreshBatch b) {
    const fd_lora_refresh_desc& p = b.d[blockIdx.y];
    const int nd = p.rp * p.K, nu = p.N * p.rp;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nd + nu; i += gridDim.x * blockDim.x) {
        if (i < nd) {
            const int j = i / p.K, k = i - j * p.K;
            const f16 v = (f16)(j < p.r ? p.down[(int64_t)j * p.K + k] : 0.f);
            ((f16*)p.d16)[(int64_t)j * p.ld_d16 + k] = v;
            ((f16*)p.dT16)[(int64_t)k * p.ld_dT16 + j] = v;
        } else {
            const int e = i - nd;
            const int n = e / p.rp, j = e - n * p.rp;
            const f16 v = (f16)(j < p.r ? p.up[(int64_t)n * p.r + j] * p.scale : 0.f);
            ((f16*)p.u16)[(int64_t)n * p.ld_u16 + j] = v;
            ((f16*)p.uT16)[(int64_t)j * p.ld_uT16 + n] = v;
        }
    }
}

extern "C" int fd_lora_refresh_multi(const fd_lora_refresh_desc* descs, int n, void* stream) {
    FD_REQUIRE(descs && n > 0, "fd_lora_refresh_multi: no pairs");
    for (int i = 0; i < n; ++i) FD_REQUIRE_DESC(descs + i, fd_lora_refresh_desc, "fd_lora_refresh_multi");
    for (int i0 = 0; i0 < n; i0 += FD_REFRESH_MAX) {
        RefreshBatch b;
        b.n = n - i0 < FD_REFRESH_MAX ? n - i0 : FD_REFRESH_MAX;
        int64_t most = 0;
        for (int i = 0; i < b.n; ++i) {
            b.d[i] = descs[i0 + i];
            const fd_lora_refresh_desc& p = b.d[i];
            FD_REQUIRE(p.down && p.up && p.d16 && p.dT16 && p.u16 && p.uT16 && p.r > 0 && p.rp >= p.r && p.K > 0 && p.N > 0, "fd_lora_refresh_multi: bad pair %d", i0 + i);
            const int64_t e = (int64_t)p.rp * (p.K + p.N);
            most = e > most ? e : most;
        }
        int bx = (int)((most + 255) / 256);
        if (bx > 64) bx = 64;
        hipLaunchKernelGGL(lora_refresh_kernel, dim3(bx, b.n), dim3(256), 0, (hipStream_t)stream, b);
    }
    return fd_check_launch("fd_lora_refresh_multi");
}

FD_WGT_SETTER(lora)
