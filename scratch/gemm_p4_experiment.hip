// REJECTED EXPERIMENT (round 2, the pipeline the round-1 verdict prescribed) -- kept for the record, not compiled into the product library.
// Measured against gemm_big_kernel<128, 320, 4, 4> on the shapes that select it (scratch/mb_mb.py with the round-1 thresholds,
// profiles/r02_gemm_p4_four_stage_ab.txt): correct (<= 8e-4), and 0-6 % SLOWER -- conv 640->640 @32^2 161.6 vs 153.9 us, conv 1280->640 @32^2 284.7 vs
// 269.3, gemm 16384x640x2560 71.2 vs 67.8; equal at the 16^2 / 8^2 levels.  Loads three k-steps ahead do not pay for twice the barriers and
// twice the fragment-read restarts of BK = 32 (the ISA is as intended: one s_waitcnt vmcnt(4) + s_barrier per step, no vmcnt(0) in the loop).
// To rebuild: copy to csrc/ as gemm_p4.hip, add to SRCS, restore the dispatch hook in gemm.hip (git history: "four-stage experiment").
//
// 16-wave 128 x 320 MFMA GEMM / stride-1 3x3 implicit-GEMM convolution with a FOUR-stage BK = 32 operand ring and counted vmcnt waits (round 2).
//
// gemm_big_kernel<128, 320, 4, 4> (gemm.hip) double-buffers BK = 64 tiles: the loads of tile t+1 are issued right behind the barrier of tile t and
// waited for with vmcnt(0) in front of the next barrier, i.e. they get exactly one tile's MFMA phase to land.  Its ablation
// (profiles/r02_gemm_ablation.txt: conv 640->640 @32^2 full 148 us, MFMA + LDS reads alone 109, loads alone 99) shows a quarter of the kernel is the two
// halves failing to overlap.  Same LDS budget, different cut: four stages of BK = 32 (4 x 32 KB), loads issued THREE k-steps ahead, and the wait in
// front of the barrier of step kt is s_waitcnt vmcnt(4) -- "everything but the loads of steps kt+1 and kt+2" -- so nothing drains in the main loop.
// For the count to be a compile-time constant every wave issues exactly two global_load_lds per k-step: the 8 + 20 sixteen-row groups of a stage are
// padded to 32 (four groups read the zero page into a spare LDS region).  Raw s_barrier (no __syncthreads(): its fence would wait vmcnt(0)).
// Staging scheme, swizzle, MFMA arrangement and epilogues are those of the other GEMM kernels (gemm_device.h).
#include "../finetune_fair_diffusion_amd/csrc/gemm_device.h"

template <int CONV>
__global__ __launch_bounds__(1024) void gemm_p4_kernel(fd_gemm_desc p, int ntm, int ntn, int gn) {
    constexpr int BM = 128, BN = 320, WGM = 4, WGN = 4, NW = 16;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;           // 32 x 80
    constexpr int TM = WTM / 16, TN = WTN / 16;             // 2 x 5
    constexpr int NST = 4;
    constexpr int GROUP = 16 * 32;                          // halfs per 16-row group (16 rows x 32 halfs = 1 KB)
    constexpr int STAGE = 32 * GROUP;                       // 32 groups per stage: A 0..7, B 8..27, pad 28..31
    extern __shared__ __attribute__((aligned(16))) f16 smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wm = wave / WGN, wn = wave % WGN;

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

    const int lrow = lane >> 2;
    const int kchunk = ((lane & 3) ^ swz_g(lane >> 4)) * 8;
    // this wave's two groups per stage: g0 = wave (A group for wave < 8, else B group wave - 8), g1 = 16 + wave (B group 8 + wave, or pad)
    const int g0 = wave, g1 = 16 + wave;
    const bool g0_is_a = wave < 8;
    const int brow0 = g0_is_a ? 0 : (wave - 8) * 16 + lrow;         // row inside the B tile served by g0
    const int brow1 = (8 + wave) * 16 + lrow;                        // row inside the B tile served by g1 (>= 320: pad)
    const bool g1_pad = wave >= 12;
    int crow_off = 0, crow_mask = 0;
    if (CONV && g0_is_a) {
        const int m = m0 + wave * 16 + lrow;
        const int hw = p.Ho * p.Wo;
        const bool valid = m < p.M;
        const int mm = valid ? m : 0;
        const int b = mm / hw;
        const int r = mm - b * hw;
        const int oy = r / p.Wo;
        const int ox = r - oy * p.Wo;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
            if (valid && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) crow_mask |= 1 << t;
        }
        crow_off = (int)(((int64_t)(b * p.H + oy) * p.W + ox) * p.lda) + kchunk;
    }
    // the zero page's address is fetched through the GOT: pin it in SGPRs once (left to the compiler it is re-loaded, with an lgkmcnt(0) wait,
    // inside every k-step)
    const f16* zp = fd_zero_page;
    asm volatile("" : "+s"(zp));
    // exactly two global_load_lds per call, whatever the wave's role
    auto issue = [&](int kt, int slot) {
        f16* st = smem + slot * STAGE;
        const f16* srcA;
        const f16* Bp;
        int64_t lb;
        int kk;
        bool kok;
        if (CONV) {
            const int cc = kt / 9;
            const int tap = kt - cc * 9;
            const int c0 = cc << 5;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int toff = ((ky - 1) * p.W + (kx - 1)) * (int)p.lda + c0;
            srcA = ((crow_mask >> tap) & 1) ? A + (int64_t)(crow_off + toff) : zp;
            Bp = B; lb = p.ldb; kk = tap * p.Cin + c0 + kchunk; kok = true;
        } else {
            const bool seg2 = kt >= nk1;
            const f16* Ap = seg2 ? A2 : A;
            Bp = seg2 ? B2 : B;
            const int64_t la = seg2 ? p.lda2 : p.lda;
            lb = seg2 ? p.ldb2 : p.ldb;
            const int Kseg = seg2 ? p.K2 : p.K;
            kk = (seg2 ? kt - nk1 : kt) * 32 + kchunk;
            kok = kk < Kseg;
            const int m = m0 + wave * 16 + lrow;
            srcA = (kok && m < p.M) ? Ap + (int64_t)m * la + kk : zp;
        }
        const int nb0 = n0 + brow0, nb1 = n0 + brow1;
        const f16* src0 = g0_is_a ? srcA : ((kok && nb0 < p.N) ? Bp + (int64_t)nb0 * lb + kk : zp);
        const f16* src1 = (!g1_pad && kok && nb1 < p.N) ? Bp + (int64_t)nb1 * lb + kk : zp;
        glds16(src0, st + g0 * GROUP);
        glds16(src1, st + g1 * GROUP);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frag_off = l15 * 32 + ((lg ^ swz_g(l15 >> 2)) * 8);

    // prologue: three steps in flight (steps beyond nk still issue their two loads -- from the zero page -- so the count stays uniform)
    auto issue_or_pad = [&](int kt, int slot) {
        if (kt < nk) issue(kt, slot);
        else {
            glds16(zp, smem + slot * STAGE + 28 * GROUP);
            glds16(zp, smem + slot * STAGE + 29 * GROUP);
        }
    };
    issue_or_pad(0, 0);
    issue_or_pad(1, 1);
    issue_or_pad(2, 2);
    for (int kt = 0; kt < nk; ++kt) {
        const int slot = kt & 3;
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // this wave's loads of step kt have landed (steps kt+1, kt+2 stay in flight)
        __builtin_amdgcn_s_barrier();                         // ... and every other wave's; all waves are done reading slot (kt-1)&3
        issue_or_pad(kt + 3, (kt + 3) & 3);
        const f16* Ab = smem + slot * STAGE + (wm * WTM / 16) * GROUP + frag_off;
        const f16* Bb = smem + slot * STAGE + (8 + wn * WTN / 16) * GROUP + frag_off;
        f16x8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *(const f16x8*)(Ab + i * GROUP);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *(const f16x8*)(Bb + j * GROUP);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = FD_MFMA_16x16x32(bf[j], af[i], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // drain the pad loads before the stages are reused by the epilogue
    __syncthreads();

    const bool lds_epi = p.out_dtype == FD_OUT_F16 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && (!p.residual || (p.ldr & 7) == 0) &&
                         (!p.rowbias || (p.ld_rowbias & 3) == 0);
    if (p.act == FD_ACT_GEGLU) {
        gemm_epilogue_geglu_lds<TM, TN, TM>(p, acc, smem + wave * WTM * (WTN + 4), m0 + wm * WTM, n0 + wn * WTN, lane);
    } else if (lds_epi) {
        gemm_epilogue_lds<TM, TN, TM>(p, acc, smem + wave * WTM * (WTN + 4), m0 + wm * WTM, n0 + wn * WTN, lane, 0, 0);
    } else {
        gemm_epilogue<TM, TN>(p, acc, m0 + wm * WTM, n0 + wn * WTN, l15, lg, 0, 0);
    }
}

static constexpr size_t P4_LDS = (size_t)4 * 32 * 16 * 32 * sizeof(f16);      // 128 KB

bool fd_gemm_p4_eligible(const fd_gemm_desc& d) {
    if (d.batch > 1 || (d.N % 320) != 0) return false;
    if (d.conv) return d.conv_mode == FD_CONV_NORMAL && (d.Cin & 31) == 0 && d.K2 == 0;
    return (d.K & 7) == 0;
}

int fd_gemm_launch_p4(const fd_gemm_desc& d, hipStream_t s) {
    const int ntm = (d.M + 127) / 128, ntn = d.N / 320;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)gemm_p4_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P4_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_p4_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P4_LDS);
    });
    const long l2_budget = 3 * 1024 * 1024;
    const long ktot = (long)d.K + d.K2;
    long gnl = l2_budget / ((long)320 * ktot * 2);
    const int gn = (int)(gnl < 1 ? 1 : (gnl > ntn ? ntn : gnl));
    if (d.conv) hipLaunchKernelGGL(gemm_p4_kernel<1>, dim3(ntm * ntn), dim3(1024), P4_LDS, s, d, ntm, ntn, gn);
    else hipLaunchKernelGGL(gemm_p4_kernel<0>, dim3(ntm * ntn), dim3(1024), P4_LDS, s, d, ntm, ntn, gn);
    return fd_check_launch("fd_gemm(p4)");
}
