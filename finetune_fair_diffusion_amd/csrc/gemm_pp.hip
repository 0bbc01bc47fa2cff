// "Ping-pong" MFMA GEMM / stride-1 3x3 implicit-GEMM convolution for gfx950 (round 3): 256 x 320 or 128 x 320 tile, 8 waves, BK = 32,
// four-stage operand ring with counted vmcnt, and the two halves of the workgroup running HALF A K-STEP APART.
//
// Why.  gemm_big_kernel<256, 320, 2, 4> (gemm.hip) keeps its eight waves in lockstep: every k-tile all of them drain vmcnt(0), meet at one
// barrier, issue the next tile's global_load_lds and restart their fragment reads AT THE SAME TIME, so on each SIMD both resident waves do
// their non-matrix work together and the matrix pipe idles meanwhile (profiles/r02_gemm_ablation.txt: conv 320 -> 320 @64^2 = 121 us full,
// 94 us with MFMAs + fragment reads only, 66 us loads only; 44 % of wave-cycles parked on s_waitcnt / barriers,
// profiles/r02_pmc_sq_wave_cycle_breakdown.txt).  Round 2 measured two coarser restructurings -- a four-stage ring with all waves in
// lockstep, and two wave groups alternating "read everything" / "multiply everything" -- and both lost.  The guide's 8-phase GEMM gets its
// speed from neither depth nor phase count but from the two waves of a SIMD being in DIFFERENT roles (MI355X_MICROARCH.md "Two waves per
// SIMD"): one issues MFMAs while the other does its waits, DMA issue and first LDS reads.
//
// How.  Waves 0-3 (rows 0-127 of the tile, group G0) and waves 4-7 (rows 128-255, G1) sit pairwise on the four SIMDs (wave w and w + 4
// share one).  Both groups run the pinned fragment-streaming loop of gemm_device.h (B resident, A two fragments ahead) over k-steps of 32,
// but G0 crosses the workgroup barrier in the MIDDLE of its k-step (fragments already in flight: it goes straight on multiplying) while G1
// crosses it at its k-step BOUNDARY (wait, DMA issue, restart of the fragment reads) -- and G0's own boundary falls in the middle of the
// interval, where G1 is in mid-stream.  One barrier per k-step as in the lockstep ring, but at every moment one wave of each SIMD is in its
// MFMA stream.
//
//   interval i (between barriers B_i and B_i+1):   G1:  issue L_i+3 | H1(i) H2(i)                        | wait vmcnt -> B_i+1
//                                                  G0:  H2(i)       | issue L_i+3 | H1(i+1)              | wait vmcnt -> B_i+1
//   L_s = the global_load_lds of k-step s into ring slot s & 3; H1 / H2 = first / second half of a k-step's MFMAs.
//
// Ring safety (4 slots).  Slot s & 3 is read by G0 from mid-interval s-1 to mid-interval s and by G1 during interval s.  L_s+3 overwrites the
// slot of step s-1: G1 issues it after B_s (it finished step s-1 before arriving there; G0 finished it in interval s-1), G0 issues it after
// its H2(s) (both groups are past B_s).  Every wave waits, in front of B_s, until only its L_s+2 are outstanding (counted vmcnt: 5 loads
// per step for waves 0-3, 4 for waves 4-7 -- 36 sixteen-row groups per step), so once B_s is crossed ALL of step s+1's operands have
// landed -- G0 starts reading them half an interval later, G1 a whole one.  (128 x 320: 28 groups per step, 4 / 3 loads per wave.)  LDS-DMA data is only ever read behind a counted vmcnt AND a
// barrier every wave has passed.  Loads past the last k-step go from the zero page into a dump area, which keeps the counts uniform.
// Raw s_barrier: __syncthreads() would fence with vmcnt(0) and drain the ring.
//
// Staging image, XOR slot permutation, zero page, swapped-operand v_mfma_f32_16x16x32 arrangement, second K-slab (LoRA rank update) and the
// LDS-staged epilogues are those of the other GEMM kernels (gemm_device.h).
#include "gemm_pp_device.h"

// CV = 2 / 3: variants 0 / 1 (dense / stride-1 3x3 gather) with the GroupNorm-statistics epilogue (fd_gemm_desc.gn_stats)
constexpr uint32_t PP_OOR = 0xFFFFFFF0u;          // a byte offset beyond every buffer: the range check of buffer_load returns zeros for the lane
template <int BM, int CV, bool PRIO>
__global__ __launch_bounds__(512) void gemm_pp_kernel(fd_gemm_desc p, int ntm, int ntn, int gn) {
    FD_WG_TRACE(5);
    constexpr int CONV = CV & 1;
    constexpr bool WSTATS = CV >= 2;
    constexpr int WTM = BM / 2, WTN = 80, TM = WTM / 16, TN = 5;
    constexpr int NGA = BM / 16, NAW = NGA / PP_NW;      // A groups per k-step, per wave (2 or 1)
    constexpr int STAGE = pp_stage<BM>();
    extern __shared__ __attribute__((aligned(16))) f16 smem[];
    const f16* zp = fd_zero_page;        // GOT load pinned in SGPRs (see gemm.hip)
    asm volatile("" : "+s"(zp));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;
    const bool lead = wave < 4;          // G0: crosses the barriers in mid-step

    int tile = xcd_remap(blockIdx.x, gridDim.x);
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
    const int m0 = mt * BM, n0 = nt * PP_BN;

    const f16* A = (const f16*)p.A;
    const f16* B = (const f16*)p.B;
    const int nk1 = (p.K + 31) >> 5, nk2 = (p.K2 + 31) >> 5, nkt = nk1 + nk2;
    // split-K: blockIdx.y owns the k-steps [kbeg, kend) and writes raw fp32 partials to the workspace
    const int nsplit = gridDim.y;
    const int kbeg = (int)((int64_t)nkt * blockIdx.y / nsplit), kend = (int)((int64_t)nkt * (blockIdx.y + 1) / nsplit);
    const int nk = kend - kbeg;

    // this lane's slot in a 16-row group: row lane >> 2, 16-byte slot lane & 3 holding k-chunk (slot ^ G[row >> 2])  (gemm_glds_kernel's image)
    const int lrow = lane >> 2;
    const int kchunk = ((lane & 3) ^ swz_g(lane >> 4)) * 8;
    // wave w stages A groups w (+ 8) and B groups w, w + 8 (, w + 16 for w < 4): 5 / 4 loads per step at BM = 256, 4 / 3 at BM = 128
    int a_off[NAW], a_mask[NAW];         // dense: row index (or -1); conv: centre-pixel offset + 9-bit tap mask
#pragma unroll
    for (int i = 0; i < NAW; ++i) {
        const int m = m0 + (wave + 8 * i) * 16 + lrow;
        const bool valid = m < p.M;
        if (CONV) {
            const int hw = p.Ho * p.Wo;
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
            a_off[i] = (int)(((int64_t)(b * p.H + oy) * p.W + ox) * p.lda) + kchunk;     // < 2^31 elements (checked by fd_gemm)
            a_mask[i] = mask;
        } else {
            a_off[i] = valid ? m : -1;
            a_mask[i] = 0;
        }
    }
    int b_row[3];                        // B row index n (or -1)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int n = n0 + (wave + 8 * i) * 16 + lrow;
        b_row[i] = (n < p.N && (wave + 8 * i) < PP_NGB) ? n : -1;
    }
    f16* const dump = smem + PP_NST * STAGE + wave * PP_GROUP;
#if __HIP_DEVICE_COMPILE__
    // Dense operands travel by buffer_load ... lds (round 6, as in gemm_halo.hip): a 32-bit per-lane byte offset fixed for the whole launch + a scalar k offset per
    // step, range-checked by the hardware -- rows beyond M / N carry an out-of-range offset and read zeros; no 64-bit address arithmetic and no pointer select per
    // piece (~8 VALU each before).  K tails (K % 32 != 0: the LoRA slab) switch the tail lanes to the out-of-range offset for that step.
    __amdgpu_buffer_rsrc_t rA, rB, rA2, rB2;
    uint32_t a_vo[NAW], a_vo2[NAW], b_vo[3], b_vo2[3];
    if (!CONV) {
        auto rsrc = [](const void* base, int64_t rows, int64_t ld, int k) {
            const int64_t bytes = base ? ((rows - 1) * ld + k) * 2 : 0;
            return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(uint32_t)bytes, 0x00020000);
        };
        rA = rsrc(p.A, p.M, p.lda, p.K); rB = rsrc(p.B, p.N, p.ldb, p.K);
        rA2 = rsrc(p.K2 ? p.A2 : nullptr, p.M, p.lda2, p.K2); rB2 = rsrc(p.K2 ? p.B2 : nullptr, p.N, p.ldb2, p.K2);
#pragma unroll
        for (int i = 0; i < NAW; ++i) {
            a_vo[i] = a_off[i] >= 0 ? (uint32_t)(a_off[i] * (int)p.lda + kchunk) * 2 : PP_OOR;
            a_vo2[i] = a_off[i] >= 0 ? (uint32_t)(a_off[i] * (int)p.lda2 + kchunk) * 2 : PP_OOR;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            b_vo[i] = b_row[i] >= 0 ? (uint32_t)(b_row[i] * (int)p.ldb + kchunk) * 2 : PP_OOR;
            b_vo2[i] = b_row[i] >= 0 ? (uint32_t)(b_row[i] * (int)p.ldb2 + kchunk) * 2 : PP_OOR;
        }
    }
#endif

    // NL global_load_lds per call: NAW A groups + (NL - NAW) B groups, for local k-step j (global step kbeg + j) into ring slot j & 3
    auto issue = [&](int j, auto nl_c) {
        constexpr int NL = decltype(nl_c)::value;
        if (j >= nk || (FD_DBG_IS(p, 1) && j >= 3)) {     // past the last k-step (or FD_GEMM_DBG = 1): keep the per-step load count uniform
#pragma unroll
            for (int i = 0; i < NL; ++i) glds16(zp, dump);
            return;
        }
        const int kt = kbeg + j;
        f16* st = smem + (j & 3) * STAGE;
        if (CONV) {
            // k order = (32-channel chunk, tap): the 9 taps of a chunk re-read the same lines shifted by a pixel
            const int cc = kt / 9;
            const int tap = kt - cc * 9;
            const int c0 = cc << 5;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int toff = ((ky - 1) * p.W + (kx - 1)) * (int)p.lda + c0;
#pragma unroll
            for (int i = 0; i < NAW; ++i) {
                const f16* src = ((a_mask[i] >> tap) & 1) ? A + (int64_t)(a_off[i] + toff) : zp;
                glds16(src, st + (wave + 8 * i) * PP_GROUP);
            }
            const int kk = tap * p.Cin + c0 + kchunk;
#pragma unroll
            for (int i = 0; i < NL - NAW; ++i) {
                const f16* src = b_row[i] >= 0 ? B + (int64_t)b_row[i] * p.ldb + kk : zp;
                glds16(src, st + (NGA + wave + 8 * i) * PP_GROUP);
            }
        } else {
#if __HIP_DEVICE_COMPILE__
            const bool seg2 = kt >= nk1;
            const int k0 = (seg2 ? kt - nk1 : kt) * 32;
            const int rem = (seg2 ? p.K2 : p.K) - k0;                       // >= 32 except in a slab's tail step
            const bool kok = kchunk < rem;
            const uint32_t so = (uint32_t)k0 * 2;
            if (!seg2) {
#pragma unroll
                for (int i = 0; i < NAW; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(st + (wave + 8 * i) * PP_GROUP), 16, kok ? a_vo[i] : PP_OOR, so, 0, 0);
#pragma unroll
                for (int i = 0; i < NL - NAW; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(st + (NGA + wave + 8 * i) * PP_GROUP), 16, kok ? b_vo[i] : PP_OOR, so, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < NAW; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA2, (__attribute__((address_space(3))) void*)(st + (wave + 8 * i) * PP_GROUP), 16, kok ? a_vo2[i] : PP_OOR, so, 0, 0);
#pragma unroll
                for (int i = 0; i < NL - NAW; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rB2, (__attribute__((address_space(3))) void*)(st + (NGA + wave + 8 * i) * PP_GROUP), 16, kok ? b_vo2[i] : PP_OOR, so, 0, 0);
            }
#endif
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offset inside a 16-row group: row l15, slot (lg ^ G[l15 >> 2])
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const uint32_t frag = (uint32_t)(l15 * 32 + ((lg ^ swz_g(l15 >> 2)) * 8)) * 2;
    const uint32_t a_frag = lds0 + (uint32_t)(wm * (WTM / 16) * PP_GROUP) * 2 + frag;
    const uint32_t b_frag = lds0 + (uint32_t)((NGA + wn * (WTN / 16)) * PP_GROUP) * 2 + frag;
    constexpr uint32_t STAGE_B = STAGE * 2, GROUP_B = PP_GROUP * 2;

    // Main loop.  ABL / NOSYNC exist in measurement builds only (FD_GEMM_DBG, scratch/mb_pp_ablate.py): 2 = no MFMAs, 6 = no fragment reads, 7 = no vmcnt waits /
    // barriers (timing only: the results are garbage); FD_GEMM_DBG = 1 sends every DMA behind the prologue to the zero page / dump group (same operation
    // counts, no memory traffic).
    auto main_loop = [&](auto abl_c, auto nosync_c) {
        constexpr int ABL = decltype(abl_c)::value;
        constexpr bool NOSYNC = decltype(nosync_c)::value;
        if (lead) {
            constexpr int NL = NAW + 3;
            std::integral_constant<int, NL> nl;
            issue(0, nl); issue(1, nl); issue(2, nl);
            wait_vm<2 * NL>();               // L_0 landed
            raw_barrier();                   // B_-1
            for (int i = 0; i < nk; ++i) {
                const uint32_t so = (uint32_t)(i & 3) * STAGE_B;
                auto mid = [&] {
                    if (!NOSYNC) {
                        wait_vm<NL>();       // this wave's L_i+1 landed (L_i+2 stays in flight)
                        raw_barrier();       // B_i, crossed in mid-step
                    }
                };
                mma_k32_mid<TM, TN, 2, GROUP_B, PRIO, decltype(mid)&, ABL>(acc, a_frag + so, b_frag + so, mid);
                issue(i + 3, nl);            // into the slot of step i-1: every wave is past B_i, i.e. done with it
            }
        } else {
            constexpr int NL = NAW + 2;
            std::integral_constant<int, NL> nl;
            issue(0, nl); issue(1, nl); issue(2, nl);
            wait_vm<2 * NL>();
            raw_barrier();                   // B_-1
            for (int i = 0; i < nk; ++i) {
                if (!NOSYNC) {
                    wait_vm<NL>();
                    raw_barrier();           // B_i, crossed at the step boundary
                }
                issue(i + 3, nl);
                const uint32_t so = (uint32_t)(i & 3) * STAGE_B;
                auto mid = [] {};
                mma_k32_mid<TM, TN, 2, GROUP_B, PRIO, decltype(mid)&, ABL>(acc, a_frag + so, b_frag + so, mid);
            }
        }
    };
#ifdef FD_BENCH_HOOKS
    if (FD_DBG_IS(p, 2)) main_loop(std::integral_constant<int, 2>{}, std::false_type{});
    else if (FD_DBG_IS(p, 6)) main_loop(std::integral_constant<int, 6>{}, std::false_type{});
    else if (FD_DBG_IS(p, 7)) main_loop(std::integral_constant<int, 0>{}, std::true_type{});
    else
#endif
    main_loop(std::integral_constant<int, 0>{}, std::false_type{});
    if (nsplit > 1) {                    // raw fp32 partials; splitk_reduce_kernel (gemm.hip) sums the slabs in a fixed order and applies the epilogue
        float* ws = (float*)p.workspace + (int64_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + l15;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + lg * 4;
                if (n < p.N) *(f32x4*)(ws + (int64_t)m * p.N + n) = acc[i][j];
            }
        }
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // drain the pad loads before the ring is reused by the epilogue
    __syncthreads();
    if (FD_DBG_IS(p, 3)) {               // measurement only (FD_GEMM_DBG = 3): no epilogue
        if (acc[0][0][0] == 12345.678f) ((f16*)p.C)[0] = (f16)1.f;
        return;
    }

    const bool lds_epi = p.out_dtype == FD_OUT_F16 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && (!p.residual || (p.ldr & 7) == 0) &&
                         (!p.rowbias || (p.ld_rowbias & 3) == 0);
    constexpr int TMC = BM == 256 ? TM / 2 : TM;             // 8 waves x 64 rows x 84 halfs = 84 KB of staging per pass
    static_assert(PP_NW * TMC * 16 * (WTN + 4) <= PP_NST * STAGE, "epilogue staging does not fit the ring");
    if (p.act == FD_ACT_GEGLU) {
        gemm_epilogue_geglu_lds<TM, TN, TMC>(p, acc, smem + wave * (TMC * 16) * (WTN + 4), m0 + wm * WTM, n0 + wn * WTN, lane);
    } else if (lds_epi) {
        gemm_epilogue_lds<TM, TN, TMC, WSTATS>(p, acc, smem + wave * (TMC * 16) * (WTN + 4), m0 + wm * WTM, n0 + wn * WTN, lane, 0, 0);
    } else {
        gemm_epilogue<TM, TN>(p, acc, m0 + wm * WTM, n0 + wn * WTN, l15, lg, 0, 0);
    }
}

template <int BM, int CV, bool PRIO>
static void launch_pp(const fd_gemm_desc& d, hipStream_t s, int ntm, int ntn, int gn, int nsplit) {
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<BM, CV, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp_lds<BM>());
    });
    hipLaunchKernelGGL((gemm_pp_kernel<BM, CV, PRIO>), dim3(ntm * ntn, nsplit), dim3(512), pp_lds<BM>(), s, d, ntm, ntn, gn);
}

template <int BM>
static void launch_pp_bm(const fd_gemm_desc& d, hipStream_t s, bool prio, int nsplit) {
    const int ntm = (d.M + BM - 1) / BM, ntn = d.N / PP_BN;
    const long l2_budget = 3 * 1024 * 1024;
    const long ktot = ((long)d.K + d.K2) / nsplit;
    long gnl = l2_budget / ((long)PP_BN * ktot * 2);
    const int gn = (int)(gnl < 1 ? 1 : (gnl > ntn ? ntn : gnl));
    if (d.gn_stats && nsplit == 1) {     // statistics-epilogue instantiations
        if (d.conv) { if (prio) launch_pp<BM, 3, true>(d, s, ntm, ntn, gn, 1); else launch_pp<BM, 3, false>(d, s, ntm, ntn, gn, 1); }
        else { if (prio) launch_pp<BM, 2, true>(d, s, ntm, ntn, gn, 1); else launch_pp<BM, 2, false>(d, s, ntm, ntn, gn, 1); }
        return;
    }
    if (d.conv) { if (prio) launch_pp<BM, 1, true>(d, s, ntm, ntn, gn, nsplit); else launch_pp<BM, 1, false>(d, s, ntm, ntn, gn, nsplit); }
    else { if (prio) launch_pp<BM, 0, true>(d, s, ntm, ntn, gn, nsplit); else launch_pp<BM, 0, false>(d, s, ntm, ntn, gn, nsplit); }
}

bool fd_gemm_pp_eligible(const fd_gemm_desc& d) {
    if (d.batch > 1 || (d.N % 320) != 0) return false;
    if (d.conv) return d.conv_mode == FD_CONV_NORMAL && (d.Cin & 31) == 0 && d.K2 == 0;
    // dense operands go through buffer descriptors with 32-bit byte offsets
    const int64_t lim = 1LL << 31;
    return (d.K & 7) == 0 && (int64_t)d.M * d.lda < lim && (int64_t)d.N * d.ldb < lim && (d.K2 == 0 || ((int64_t)d.M * d.lda2 < lim && (int64_t)d.N * d.ldb2 < lim));
}

// gemm_halo.hip: the same loop with the A operand of a stride-1 3x3 convolution staged once per channel chunk (round 6)
bool fd_conv_halo_eligible(const fd_gemm_desc& d, int bm);
int fd_conv_halo_launch(const fd_gemm_desc& d, hipStream_t s, bool prio, int bm);
bool fd_conv_halo_takes(const fd_gemm_desc& d, int bm, int nsplit) {
#ifdef FD_BENCH_HOOKS
    const char* e = getenv("FD_CONV_HALO");          // measurement build: FD_CONV_HALO=0 keeps the per-tap gather (re-read on every call for in-process A/Bs)
    if (e && atoi(e) == 0) return false;
#endif
    return nsplit == 1 && fd_conv_halo_eligible(d, bm);
}

// bm: 256 or 128 rows per tile; nsplit > 1: split-K partials into d.workspace (the caller launches the reduction)
int fd_gemm_launch_pp(const fd_gemm_desc& d, hipStream_t s, bool prio, int bm, int nsplit) {
    if (fd_conv_halo_takes(d, bm, nsplit)) return fd_conv_halo_launch(d, s, prio, bm);
    if (bm == 256) launch_pp_bm<256>(d, s, prio, nsplit);
    else launch_pp_bm<128>(d, s, prio, nsplit);
    return fd_check_launch("fd_gemm(pp)");
}

FD_WGT_SETTER(gemm_pp)
