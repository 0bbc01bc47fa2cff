// Persistent ("streaming") form of the 128 x 320 ping-pong GEMM for the SHORT-K dense projections of the 64^2 / 32^2 levels (round 3).
//
// Why.  K = 320 ... 1280 means 10-40 k-steps of 32 per tile; with one workgroup per CU (the operand ring fills the LDS) the prologue (first
// loads' latency), the k-loop and the epilogue (accumulators -> LDS -> 128-byte row segments -> HBM) of a tile run one after the other and
// nothing else is resident to fill the gaps: profiles/r02_gemm_ablation.txt prices the exposed epilogue at 32-35 % of the kernel on
// 65536 x 320 x 320 and 65536 x 2560 x 320.  Here a workgroup owns a contiguous RANGE of tiles and never stops streaming:
//   * the operand ring (gemm_pp.hip: BK = 32, four slots, loads three k-steps ahead, counted vmcnt, the two wave groups half a k-step
//     apart) runs ACROSS tile boundaries -- the first three k-steps of tile t+1 are in flight while tile t finishes;
//   * the epilogue has its own wave-private LDS region (32 rows x 80 columns per wave, two passes), so it needs no workgroup barrier and
//     does not touch the ring: while a wave converts and stores its 64 x 80 block, the other wave of its SIMD (the other group) multiplies;
//   * the epilogue's global stores are never waited for: CDNA4's vmcnt counts stores in issue order with the loads, so the two k-step waits
//     that follow an epilogue allow its NS stores to stay outstanding (vmcnt(NL + NS) instead of vmcnt(NL)); they drain under the next
//     tile's first k-steps.  Bias vectors live in registers per column block (reloaded only when the range crosses into the next column
//     block), residual rows are fetched at the start of the epilogue in one batch.
// Tiles are ordered row-block-fastest inside a column block, so a workgroup's B tile (320 x K) and bias stay put for its whole range and
// workgroups w, w + 32, ... (same XCD) read the same A row blocks of different column blocks at about the same time.
//
// Ring / barrier protocol, staging image, fragment schedule: gemm_pp.hip.  Epilogue modes: bias (+ residual), or GEGLU (FF1 of forwards
// that do not record the pre-gate projection).  Everything else stays on the one-tile-per-workgroup kernels (fd_gemm_pps_eligible).
#include "gemm_pp_device.h"

constexpr int PPS_BM = 128;
constexpr int PPS_EPI_ROWS = 32;                                         // rows of a wave's 64 x 80 block staged per pass
constexpr int PPS_EPI_HALFS = PPS_EPI_ROWS * (80 + 4);                   // per wave
constexpr size_t pps_lds() { return (size_t)(PP_NST * pp_stage<PPS_BM>() + PP_GROUP + PP_NW * PPS_EPI_HALFS) * sizeof(f16); }   // 158,720 B

// bias (+ residual) epilogue of one wave's 64 x 80 block: 12 store instructions, no vmcnt wait after the first store
template <int TM, int TN>
static __device__ __forceinline__ void pps_epilogue_plain(const fd_gemm_desc& p, f32x4 (&acc)[TM][TN], const f32x4 (&bv)[TN], f16* wave_lds,
                                                          int mbase, int nbase, int lane, const f16* zp) {
    constexpr int TMC = PPS_EPI_ROWS / 16, WTN = TN * 16, LDW = WTN + 4, CPR = WTN / 8, RPI = 64 / CPR, NI = (PPS_EPI_ROWS + RPI - 1) / RPI;
    const int l15 = lane & 15, lg = lane >> 4;
    const int cr = lane / CPR, cc = (lane % CPR) * 8;
    const f16* R = (const f16*)p.residual;
    static_assert(TM / TMC == 2 && NI == 6, "the wait statement below names 12 residual vectors");
    f16x8 rv[TM / TMC][NI];
    if (R) {
        // All residual rows of the block up front (a load issued behind a store would wait for that store: vmcnt is in issue order), as
        // loads hipcc does not count: a counted load that some lanes skip leaves "maybe pending" state that hipcc drains with vmcnt(0)
        // inside the k-loop.  Lanes without a row read the zero page, so every wave issues exactly 12.
#pragma unroll
        for (int c = 0; c < TM / TMC; ++c)
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int row = k * RPI + cr, m = mbase + c * PPS_EPI_ROWS + row;
                const f16* src = (cr < RPI && row < PPS_EPI_ROWS && m < p.M) ? R + (int64_t)m * p.ldr + nbase + cc : zp;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rv[c][k]) : "v"(src) : "memory");
            }
    }
#pragma unroll
    for (int c = 0; c < TM / TMC; ++c) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int ii = 0; ii < TMC; ++ii) {
                const f32x4 v = acc[c * TMC + ii][j] + bv[j];
                *(f16x4*)(wave_lds + (ii * 16 + l15) * LDW + j * 16 + lg * 4) = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (R && c == 0)         // the residual vectors have landed (they are the youngest vector-memory operations: a full drain)
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(rv[0][0]), "+v"(rv[0][1]), "+v"(rv[0][2]), "+v"(rv[0][3]), "+v"(rv[0][4]), "+v"(rv[0][5]), "+v"(rv[1][0]),
                           "+v"(rv[1][1]), "+v"(rv[1][2]), "+v"(rv[1][3]), "+v"(rv[1][4]), "+v"(rv[1][5])
                         :
                         : "memory");
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int row = k * RPI + cr, m = mbase + c * PPS_EPI_ROWS + row;
            if (cr < RPI && row < PPS_EPI_ROWS && m < p.M) {
                const f16x4 lo = *(const f16x4*)(wave_lds + row * LDW + cc);
                const f16x4 hi = *(const f16x4*)(wave_lds + row * LDW + cc + 4);
                f16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (R) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] + (float)rv[c][k][e]);
                }
                *(f16x8*)((f16*)p.C + (int64_t)m * p.ldc + nbase + cc) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's staging reads have returned before the rows are rewritten
        __builtin_amdgcn_wave_barrier();
    }
}

// GEGLU epilogue (gemm_epilogue_geglu_lds without the pre-gate output): B rows interleave (value_c, gate_c); 6 store instructions
template <int TM, int TN>
static __device__ __forceinline__ void pps_epilogue_geglu(const fd_gemm_desc& p, f32x4 (&acc)[TM][TN], const f32x4 (&bv)[TN], f16* wave_lds,
                                                          int mbase, int nbase, int lane) {
    constexpr int TMC = PPS_EPI_ROWS / 16, WTO = TN * 8, LDW = WTO + 4, CPR = WTO / 8, RPI = 64 / CPR, NI = (PPS_EPI_ROWS + RPI - 1) / RPI;
    const int l15 = lane & 15, lg = lane >> 4;
    const int cr = lane / CPR, cc = (lane % CPR) * 8;
#pragma unroll
    for (int c = 0; c < TM / TMC; ++c) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int ii = 0; ii < TMC; ++ii) {
                const f32x4 v = acc[c * TMC + ii][j] + bv[j];
                const f16 v0 = (f16)v[0], g0 = (f16)v[1], v1 = (f16)v[2], g1 = (f16)v[3];
                *(f16x2*)(wave_lds + (ii * 16 + l15) * LDW + j * 8 + lg * 2) =
                    (f16x2){(f16)((float)v0 * gelu_erf_f((float)g0)), (f16)((float)v1 * gelu_erf_f((float)g1))};
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int row = k * RPI + cr, m = mbase + c * PPS_EPI_ROWS + row;
            if (cr < RPI && row < PPS_EPI_ROWS && m < p.M) {
                const f16x4 lo = *(const f16x4*)(wave_lds + row * LDW + cc);
                const f16x4 hi = *(const f16x4*)(wave_lds + row * LDW + cc + 4);
                *(f16x8*)((f16*)p.C + (int64_t)m * p.ldc + (nbase >> 1) + cc) = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int EPI>   // 0: bias (+ residual), 1: GEGLU
__global__ __launch_bounds__(512) void gemm_pps_kernel(fd_gemm_desc p, int ntm, int ntn) {
    constexpr int BM = PPS_BM, WTM = BM / 2, WTN = 80, TM = WTM / 16, TN = 5;
    constexpr int NGA = BM / 16;                         // 8 A groups per k-step: one per wave
    constexpr int STAGE = pp_stage<BM>();
    constexpr int NS = EPI == 0 ? 12 : 6;                // store instructions of one epilogue (full tile)
    extern __shared__ __attribute__((aligned(16))) f16 smem[];
    const f16* zp = fd_zero_page;
    asm volatile("" : "+s"(zp));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;
    const bool lead = wave < 4;

    // this workgroup's tile range; tile L = column block L / ntm, row block L % ntm
    const int ntiles = ntm * ntn, G = gridDim.x, w = blockIdx.x;
    const int base = ntiles / G, rem = ntiles - base * G;
    const int Lbeg = w * base + min(w, rem), Lend = Lbeg + base + (w < rem ? 1 : 0);

    const f16* A = (const f16*)p.A;
    const f16* B = (const f16*)p.B;
    const f16* A2 = (const f16*)p.A2;
    const f16* B2 = (const f16*)p.B2;
    const int nk1 = (p.K + 31) >> 5, nk2 = (p.K2 + 31) >> 5, nkt = nk1 + nk2;

    const int lrow = lane >> 2;
    const int kchunk = ((lane & 3) ^ swz_g(lane >> 4)) * 8;
    f16* const dump = smem + PP_NST * STAGE;             // one group, shared: never read
    f16* const wave_lds = dump + PP_GROUP + wave * PPS_EPI_HALFS;

    // ---- issue side: runs three k-steps ahead of the multiplies, so it crosses into the next tile first
    int iL = Lbeg, ik = 0, ig = 0;
    int i_arow = -1, i_brow[3] = {-1, -1, -1};
    auto set_rows = [&](int L) {
        const int nt = L / ntm, mt = L - nt * ntm;
        const int m = mt * BM + wave * 16 + lrow;
        i_arow = (L < Lend && m < p.M) ? m : -1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int n = nt * PP_BN + (wave + 8 * i) * 16 + lrow;
            i_brow[i] = (L < Lend && (wave + 8 * i) < PP_NGB && n < p.N) ? n : -1;
        }
    };
    set_rows(iL);
    auto issue = [&](auto nl_c) {
        constexpr int NL = decltype(nl_c)::value;        // 1 A group + (NL - 1) B groups per wave and k-step
        f16* st = smem + (ig & 3) * STAGE;
        ++ig;
        if (iL >= Lend) {                                // past the last tile: keep the per-step load count uniform
#pragma unroll
            for (int i = 0; i < NL; ++i) glds16(zp, dump);
            return;
        }
        const bool seg2 = ik >= nk1;
        const f16* Ap = seg2 ? A2 : A;
        const f16* Bp = seg2 ? B2 : B;
        const int64_t la = seg2 ? p.lda2 : p.lda, lb = seg2 ? p.ldb2 : p.ldb;
        const int Kseg = seg2 ? p.K2 : p.K;
        const int kk = (seg2 ? ik - nk1 : ik) * 32 + kchunk;
        const bool kok = kk < Kseg;
        glds16((kok && i_arow >= 0) ? Ap + (int64_t)i_arow * la + kk : zp, st + wave * PP_GROUP);
#pragma unroll
        for (int i = 0; i < NL - 1; ++i)
            glds16((kok && i_brow[i] >= 0) ? Bp + (int64_t)i_brow[i] * lb + kk : zp, st + (NGA + wave + 8 * i) * PP_GROUP);
        if (++ik == nkt) {
            ik = 0;
            ++iL;
            set_rows(iL);
        }
    };

    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const uint32_t frag = (uint32_t)(l15 * 32 + ((lg ^ swz_g(l15 >> 2)) * 8)) * 2;
    const uint32_t a_frag = lds0 + (uint32_t)(wm * (WTM / 16) * PP_GROUP) * 2 + frag;
    const uint32_t b_frag = lds0 + (uint32_t)((NGA + wn * (WTN / 16)) * PP_GROUP) * 2 + frag;
    constexpr uint32_t STAGE_B = STAGE * 2, GROUP_B = PP_GROUP * 2;

    f32x4 acc[TM][TN];
    f32x4 bv[TN];
    int bias_nt = -1;
    auto tile_epilogue = [&](int L) -> int {             // returns how many of the following k-step waits may leave NS stores outstanding
        const int nt = L / ntm, mt = L - nt * ntm;
        const int m0 = mt * BM, n0 = nt * PP_BN;
        if (nt != bias_nt) {                             // first tile of a column block: fetch the bias vectors and wait for them right here
            bias_nt = nt;                                //  (a full vmcnt drain, once per column block: the ring's loads land with it)
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p.bias) {                                // uncounted loads + one wait statement naming them (see pps_epilogue_plain)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv[j]) : "v"(p.bias + n0 + wn * WTN + j * 16 + lg * 4) : "memory");
                static_assert(TN == 5, "the wait statement names 5 bias vectors");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]) : : "memory");
            }
        }
        if (EPI == 0) pps_epilogue_plain<TM, TN>(p, acc, bv, wave_lds, m0 + wm * WTM, n0 + wn * WTN, lane, zp);
        else pps_epilogue_geglu<TM, TN>(p, acc, bv, wave_lds, m0 + wm * WTM, n0 + wn * WTN, lane);
        return (m0 + BM <= p.M) ? 2 : 0;                 // ragged row blocks issue fewer stores: strict waits
    };
    auto zero_acc = [&] {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };

    int g = 0, loose = 0;
    if (lead) {
        constexpr int NL = 4;
        std::integral_constant<int, NL> nl;
        issue(nl); issue(nl); issue(nl);
        wait_vm<2 * NL>();                               // L_0 landed
        raw_barrier();                                   // B_-1
        for (int L = Lbeg; L < Lend; ++L) {
            zero_acc();
            for (int i = 0; i < nkt; ++i, ++g) {
                const uint32_t so = (uint32_t)(g & 3) * STAGE_B;
                mma_k32_mid<TM, TN, 2, GROUP_B, true>(acc, a_frag + so, b_frag + so, [&] {
                    if (loose > 0) wait_vm<NL + NS>();   // L_g+1 landed; L_g+2 and the last epilogue's stores may stay in flight
                    else wait_vm<NL>();
                    raw_barrier();                       // B_g, crossed in mid-step
                });
                loose = loose > 0 ? loose - 1 : 0;
                issue(nl);
            }
            asm volatile("" ::: "memory");
            loose = tile_epilogue(L);
            asm volatile("" ::: "memory");
        }
    } else {
        constexpr int NL = 3;
        std::integral_constant<int, NL> nl;
        issue(nl); issue(nl); issue(nl);
        wait_vm<2 * NL>();
        raw_barrier();                                   // B_-1
        for (int L = Lbeg; L < Lend; ++L) {
            zero_acc();
            for (int i = 0; i < nkt; ++i, ++g) {
                if (loose > 0) wait_vm<NL + NS>();
                else wait_vm<NL>();
                raw_barrier();                           // B_g, crossed at the step boundary
                loose = loose > 0 ? loose - 1 : 0;
                issue(nl);
                const uint32_t so = (uint32_t)(g & 3) * STAGE_B;
                mma_k32_mid<TM, TN, 2, GROUP_B, true>(acc, a_frag + so, b_frag + so, [] {});
            }
            asm volatile("" ::: "memory");
            loose = tile_epilogue(L);
            asm volatile("" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the pad loads must have landed before this workgroup's LDS is handed on
}

// Shapes the streaming kernel takes: plain fp16 dense GEMMs (optional LoRA K-slab) with a bias / residual or GEGLU epilogue whose output
// rows can be stored as 16-byte pieces.  Everything else (convolutions, row bias, activations, alpha, fp32 output, recorded pre-gate
// output, batched, ragged N) stays on the one-tile-per-workgroup kernels.
bool fd_gemm_pps_eligible(const fd_gemm_desc& d) {
    if (d.conv || d.batch > 1 || (d.N % 320) != 0 || (d.K & 7) != 0 || d.colscale_cols) return false;
    if (d.out_dtype != FD_OUT_F16 || d.rowbias || d.alpha != 1.f || (d.ldc & 7) != 0) return false;
    if (d.act == FD_ACT_GEGLU) return d.residual == nullptr;
    if (d.act != FD_ACT_NONE) return false;
    return !d.residual || (d.ldr & 7) == 0;
}

int fd_gemm_launch_pps(const fd_gemm_desc& d, hipStream_t s, int max_wg) {
    const int ntm = (d.M + PPS_BM - 1) / PPS_BM, ntn = d.N / PP_BN;
    const int ntiles = ntm * ntn, G = ntiles < max_wg ? ntiles : max_wg;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)gemm_pps_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pps_lds());
        (void)hipFuncSetAttribute((const void*)gemm_pps_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pps_lds());
    });
    if (d.act == FD_ACT_GEGLU) hipLaunchKernelGGL((gemm_pps_kernel<1>), dim3(G), dim3(512), pps_lds(), s, d, ntm, ntn);
    else hipLaunchKernelGGL((gemm_pps_kernel<0>), dim3(G), dim3(512), pps_lds(), s, d, ntm, ntn);
    return fd_check_launch("fd_gemm(pps)");
}
