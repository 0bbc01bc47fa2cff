// REJECTED EXPERIMENT (round 2) -- kept for the record, not compiled into the product library.
// Measured against the shipped tile policy on the step's shapes (scratch/mb_mb.py, profiles/r02_gemm_multiblock_ab.txt): correct (errors
// <= 7e-4) but slower on 21 of 22 shapes -- conv 640->640 @32^2 192 vs 149 us, gemm 65536x320x320 35.6 vs 32.8 us, gemm 4096x1280x5120
// 165 vs 84 us; only gemm 4096x3840x1280 wins (66 vs 72 us).  Two independent 4-wave workgroups per CU lose more in MFMA issue (one wave
// per SIMD per workgroup, 13 fragment reads per 40 MFMAs behind every barrier) than they gain in overlap.
// To rebuild: copy to csrc/, add to SRCS, restore the dispatch hook in gemm.hip (git history: "multi-block experiment").
//
// Multi-block-per-CU MFMA GEMM / stride-1 3x3 implicit-GEMM convolution for gfx950 (round 2).
//
// The 8/16-wave BK = 64 kernels of gemm.hip own a whole CU (114-147 KB of LDS, the full register file): one workgroup per CU, all of
// its waves in lockstep behind one barrier per k-tile.  Their ablation (scratch/mb_ablate.py, profiles/r02_gemm_ablation.txt) shows what
// that costs on this workload's shapes: operand loads alone and MFMAs alone each take ~70 % of the kernel and overlap only partially
// (conv 640->640 @32^2: 99 / 109 / 148 us), and the epilogue -- 10 % (long-K convs) to 35 % (FF1: 335 MB written) -- overlaps with
// nothing, because no second workgroup is resident to compute underneath it.
//
// Here a 128 x 320 x 32 tile is computed by FOUR waves (1 x 4: wave tile 128 x 80, 160 accumulator registers, two waves per SIMD), its
// two operand stages take 56 KB of LDS, so TWO independent workgroups share a CU: while one waits for its loads, sits in its barrier or
// streams its epilogue to HBM, the other one issues MFMAs.  Same staging scheme as the other kernels (global_load_lds_dwordx4 into
// unpadded 64-byte rows, XOR slot permutation, zero page for padding / tails), same swapped-operand v_mfma_f32_16x16x32 arrangement,
// same LDS-staged epilogues (bias / row-bias / activation / residual / fused GEGLU), LoRA rank update as a second K-slab.
#include "../finetune_fair_diffusion_amd/csrc/gemm_device.h"

template <int BM, int BN, int WGM, int WGN, int CONV>
__global__ __launch_bounds__(WGM * WGN * 64, 2) void gemm_mb_kernel(fd_gemm_desc p, int ntm, int ntn, int gn) {
    constexpr int NW = WGM * WGN;
    static_assert(NW == 4, "4 waves");
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int NA = BM / 16, NB = BN / 16;               // 16-row groups (one glds instruction each) per k-tile
    constexpr int AI = (NA + NW - 1) / NW, BI = (NB + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) f16 smem[];
    f16* As = smem;                      // [2][BM][32]
    f16* Bs = smem + 2 * BM * 32;        // [2][BN][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wm = wave / WGN, wn = wave % WGN;

    // tile order: XCD-contiguous ranges, n-tiles banded so that a band's B slab stays in the XCD's L2 (as gemm_big_kernel)
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int mt, nt;
    if (gn >= ntn) {
        mt = tile / ntn;
        nt = tile - mt * ntn;
    } else {
        const int per = ntm * gn, nbands = (ntn + gn - 1) / gn;
        const int band = min(tile / per, nbands - 1);
        const int r = tile - band * per;
        const int w = band == nbands - 1 ? ntn - band * gn : gn;
        mt = r / w;
        nt = band * gn + (r - mt * w);
    }
    const int m0 = mt * BM, n0 = nt * BN;

    const f16* A = (const f16*)p.A;
    const f16* B = (const f16*)p.B;
    const f16* A2 = (const f16*)p.A2;
    const f16* B2 = (const f16*)p.B2;
    const int nk1 = (p.K + 31) >> 5, nk2 = (p.K2 + 31) >> 5, nk = nk1 + nk2;

    // this lane's slot in a 16-row group: row lane>>2, 16-byte slot lane&3 holding k-chunk (slot ^ G[row>>2])
    const int lrow = lane >> 2;
    const int kchunk = ((lane & 3) ^ swz_g(lane >> 4)) * 8;
    int crow_off[AI], crow_mask[AI];
    if (CONV) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int m = m0 + (wave + i * NW) * 16 + lrow;
            const int hw = p.Ho * p.Wo;
            const bool valid = m < p.M && (wave + i * NW) < NA;
            const int mm = valid ? m : 0;
            const int b = mm / hw;
            const int r = mm - b * hw;
            const int oy = r / p.Wo;
            const int ox = r - oy * p.Wo;
            int mask = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
                if (valid && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mask |= 1 << t;
            }
            crow_off[i] = (int)(((int64_t)(b * p.H + oy) * p.W + ox) * p.lda) + kchunk;   // < 2^31 elements (checked by the launcher)
            crow_mask[i] = mask;
        }
    }
    auto issue = [&](int kt, int buf) {
        if (CONV) {
            // k order = (32-channel chunk, tap): the 9 taps of a chunk re-read the same lines shifted by a pixel (L1 / L2 hits)
            const int cc = kt / 9;
            const int tap = kt - cc * 9;
            const int c0 = cc << 5;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int toff = ((ky - 1) * p.W + (kx - 1)) * (int)p.lda + c0;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int g = wave + i * NW;
                if (g < NA) {
                    const bool ok = (crow_mask[i] >> tap) & 1;
                    const f16* src = ok ? A + (int64_t)(crow_off[i] + toff) : fd_zero_page;
                    glds16(src, As + (buf * BM + g * 16) * 32);
                }
            }
            const int kk = tap * p.Cin + c0 + kchunk;
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const int g = wave + i * NW;
                if (g < NB) {
                    const int n = n0 + g * 16 + lrow;
                    const f16* src = (n < p.N) ? B + (int64_t)n * p.ldb + kk : fd_zero_page;
                    glds16(src, Bs + (buf * BN + g * 16) * 32);
                }
            }
        } else {
            const bool seg2 = kt >= nk1;
            const f16* Ap = seg2 ? A2 : A;
            const f16* Bp = seg2 ? B2 : B;
            const int64_t la = seg2 ? p.lda2 : p.lda, lb = seg2 ? p.ldb2 : p.ldb;
            const int Kseg = seg2 ? p.K2 : p.K;
            const int kk = (seg2 ? kt - nk1 : kt) * 32 + kchunk;
            const bool kok = kk < Kseg;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int g = wave + i * NW;
                if (g < NA) {
                    const int m = m0 + g * 16 + lrow;
                    const f16* src = (kok && m < p.M) ? Ap + (int64_t)m * la + kk : fd_zero_page;
                    glds16(src, As + (buf * BM + g * 16) * 32);
                }
            }
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const int g = wave + i * NW;
                if (g < NB) {
                    const int n = n0 + g * 16 + lrow;
                    const f16* src = (kok && n < p.N) ? Bp + (int64_t)n * lb + kk : fd_zero_page;
                    glds16(src, Bs + (buf * BN + g * 16) * 32);
                }
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read inside a 16-row group (1024 bytes): row l15, slot (lg ^ G[l15>>2])
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const uint32_t frag_off = (uint32_t)(l15 * 32 + ((lg ^ swz_g(l15 >> 2)) * 8)) * 2;

    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
        const uint32_t a_base = lds0 + (uint32_t)((buf * BM + wm * WTM) * 32) * 2 + frag_off;
        const uint32_t b_base = lds0 + (uint32_t)((2 * BM + buf * BN + wn * WTN) * 32) * 2 + frag_off;
        mma_k32<TM, TN, 2, 1024>(acc, a_base, b_base);
    }

    const bool lds_epi = p.out_dtype == FD_OUT_F16 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && (!p.residual || (p.ldr & 7) == 0) &&
                         (!p.rowbias || (p.ld_rowbias & 3) == 0);
    constexpr int LDS_HALFS = 2 * (BM + BN) * 32;
    constexpr int TMC = (NW * WTM * (WTN + 4) <= LDS_HALFS) ? TM : ((NW * (WTM / 2) * (WTN + 4) <= LDS_HALFS) ? TM / 2 : TM / 4);
    static_assert(NW * TMC * 16 * (WTN + 4) <= LDS_HALFS, "epilogue staging does not fit");
    if (p.act == FD_ACT_GEGLU) {
        __syncthreads();
        gemm_epilogue_geglu_lds<TM, TN, TMC>(p, acc, smem + wave * (TMC * 16) * (WTN + 4), m0 + wm * WTM, n0 + wn * WTN, lane);
    } else if (lds_epi) {
        __syncthreads();   // every wave is done reading the operand stages before they are reused as epilogue staging
        gemm_epilogue_lds<TM, TN, TMC>(p, acc, smem + wave * (TMC * 16) * (WTN + 4), m0 + wm * WTM, n0 + wn * WTN, lane, 0, 0);
    } else {
        gemm_epilogue<TM, TN>(p, acc, m0 + wm * WTM, n0 + wn * WTN, l15, lg, 0, 0);
    }
}

template <int BM, int BN, int WGM, int WGN>
static int launch_mb_t(const fd_gemm_desc& d, hipStream_t s) {
    const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
    constexpr size_t lds = (size_t)2 * (BM + BN) * 32 * sizeof(f16);
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)gemm_mb_kernel<BM, BN, WGM, WGN, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gemm_mb_kernel<BM, BN, WGM, WGN, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    const long l2_budget = 3 * 1024 * 1024;
    const long ktot = (long)d.K + d.K2;
    long gnl = l2_budget / ((long)BN * ktot * 2);
    const int gn = (int)(gnl < 1 ? 1 : (gnl > ntn ? ntn : gnl));
    if (d.conv) hipLaunchKernelGGL((gemm_mb_kernel<BM, BN, WGM, WGN, 1>), dim3(ntm * ntn), dim3(WGM * WGN * 64), lds, s, d, ntm, ntn, gn);
    else hipLaunchKernelGGL((gemm_mb_kernel<BM, BN, WGM, WGN, 0>), dim3(ntm * ntn), dim3(WGM * WGN * 64), lds, s, d, ntm, ntn, gn);
    return fd_check_launch("fd_gemm(mb)");
}

// problems this family takes: unbatched, N a multiple of 320, K-tiles of 32 that do not straddle a conv tap, stride-1 3x3 gather only
bool fd_gemm_mb_eligible(const fd_gemm_desc& d) {
    if (d.batch > 1 || (d.N % 320) != 0) return false;
    if (d.conv) return d.conv_mode == FD_CONV_NORMAL && (d.Cin & 31) == 0 && d.K2 == 0;
    return (d.K & 7) == 0;
}

int fd_gemm_launch_mb(const fd_gemm_desc& d, hipStream_t s) { return launch_mb_t<128, 320, 1, 4>(d, s); }
