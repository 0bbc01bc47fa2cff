// "Register-B" streaming MFMA GEMM for the dense projections of the U-Net (round 5): C[M, N] = epilogue(A[M, K] . B[N, K]^T + A2 . B2^T), N % 320 == 0, K % 320 == 0.
//
// Why.  The in-situ counters of the round-4 step (profiles/r05_pmc_sq_in_situ_step.csv) show the dense kernels PARKED: gemm_big_kernel<256, 320, 4, 4, 0>
// spends 55 % of its wave-cycles in s_waitcnt / s_barrier and 15 % issuing; 16384 x 640 x 640 takes 30 us and 4096 x 1280 x 1280 38 us where both the
// matrix pipe and HBM would need 7.  Every kernel of gemm.hip / gemm_pp.hip lands BOTH operands in LDS, so the bytes a CU can have in flight are capped by
// the LDS ring (74 - 108 KB), and in the step the A operand is cold (written by the previous kernel: ~2 us away): throughput = bytes in flight / latency
// ~ 50 KB/us per CU, half of what the MFMAs of a 128 x 320 tile consume.  The register file is the bigger landing zone (512 KB per CU), and the weight
// operand needs no sharing between waves if every wave owns its own columns:
//
//   * a workgroup is FOUR waves, one per SIMD; wave w owns columns [80 w, 80 w + 80) of the 128 x 320 tile and ALL 128 rows (accumulators 8 x 5 x f32x4);
//   * B travels global -> VGPR directly in MFMA layout (lane (l15, lg) reads the 16 bytes B[n0 + 16 j + l15][32 s + 8 lg ..]), through a ring of
//     D = 10 k-steps of registers (200 VGPRs): 50 KB per wave, 200 KB per CU in flight, nothing of it in LDS, no duplicate fetch (a column slice has one
//     owner).  D = 10 divides the k-step count of every projection of the U-Net (K % 320 == 0), so ring slots are compile-time register names in a
//     ten-step unrolled loop and a tile always starts at slot 0;
//   * A travels global -> LDS with global_load_lds_dwordx4 (the 16-row x 64-byte group image and XOR slot permutation of gemm_pp.hip) through a ring of
//     stages of 8 KB, ten k-steps ahead as well; a wave multiplies fragment row i (five MFMAs) and then reads row i of the NEXT k-step into the same
//     registers -- every LDS read has a whole k-step (40 MFMAs) to return;
//   * one counted s_waitcnt vmcnt + one raw s_barrier per k-step of 32.  All vector-memory operations of a wave retire in issue order, so "the loads of the
//     next k-step have landed" is "at most 8 x 7 operations outstanding" (7 = 5 B fragments + 2 A groups per wave and step, issued ten steps ahead); anything
//     else the wave issues in between -- LoRA-slab loads, epilogue stores, bias / residual loads -- only makes that count conservative;
//   * the LoRA rank update (K2 <= 32: one k-step of t . up^T) has its own five registers and its own LDS stage, fetched at the start of the tile;
//   * workgroups are PERSISTENT: each owns a contiguous range of tiles (row blocks fastest inside a column block) and the rings run across tile
//     boundaries, so the first ten k-steps of the next tile are in flight while this tile's epilogue runs.
// Per k-step a CU moves 8 KB (A) + 20 KB (B) for 2.6 MFLOP like the 128 x 320 ping-pong tile, but with ~280 KB in flight instead of 84.
//
// Arithmetic: v_mfma_f32_16x16x32 with swapped operands and the k order of the other kernels (32-wide k blocks ascending, LoRA slab last): results are
// bit-identical to gemm_big_kernel / gemm_pp_kernel.  Epilogues: gemm_epilogue_lds / gemm_epilogue_geglu_lds of gemm_device.h on a wave-private staging area.
#include "gemm_pp_device.h"

#ifndef RB_BM_ROWS
#define RB_BM_ROWS 96
#endif
constexpr int RB_BM = RB_BM_ROWS, RB_BN = 320, RB_D = 10, RB_TM = RB_BM / 16, RB_TN = 5, RB_NW = 4;
constexpr int RB_AG = 2;                                           // A groups (16 rows) a wave issues per stage; groups beyond the tile go to a dump area
constexpr int RB_NSTA = 12;                                        // A ring stages (ten in flight, the one being read, one of slack)
constexpr int RB_STAGE = RB_BM * 32;                               // halfs per A stage: 8 groups of 16 rows x 64 bytes
constexpr int RB_NL = RB_TN + RB_AG;                  // vector-memory operations per wave and k-step: 5 B fragments + 2 A groups
// THE WAVE'S vmcnt IS A 6-BIT COUNTER AND IT WRAPS: with ten bundles of seven operations outstanding (70) the counted waits stopped meaning anything --
// products of stale registers / stages on cold operands, correct with a full drain per step, correct again as soon as at most 63 operations were ever
// outstanding (profiles/r05_gemm_rb_bringup.txt).  So the ring has ten SLOTS (static register names need the period ten) but at most RB_W + RB_NL = 49
// of its operations are in flight: the wait at the top of a step lets six bundles stay outstanding, the re-fill adds the seventh.  On top of that come
// the LoRA-slab bundle (7, behind its own guard wait) and the epilogue's stores / residual / bias operations (at most RB_EPI_OPS, behind a guard wait that
// leaves room for them): the count never exceeds 63.
constexpr int RB_LA = 9;                                           // bundles issued ahead of the step being multiplied
#ifdef RB_WAIT
constexpr int RB_W = RB_WAIT;                                      // measurement override
#else
constexpr int RB_W = (RB_LA - 2) * RB_NL;                          // at the top of step g the bundles g+2 .. g+LA-1 may stay outstanding
#endif
constexpr int RB_EPI_OPS = 34;                                     // GEGLU with the recorded pre-gate output: 3 passes x (6 + 3) stores, + 5 bias vectors (+ slack)
constexpr int RB_W_EPI = 63 - RB_EPI_OPS;                          // operations that may be outstanding when an epilogue starts
constexpr int RB_TMC = 2;                                          // 16-row groups staged per epilogue pass
constexpr int RB_EPI_HALFS = RB_TMC * 16 * (80 + 4);               // per wave
constexpr size_t rb_lds() { return (size_t)((RB_NSTA + 1) * RB_STAGE + RB_NW * RB_TN * PP_GROUP + RB_NW * RB_EPI_HALFS + PP_GROUP) * sizeof(f16) + RB_BN * sizeof(float); }      // 96 KB ring + 8 KB slab stage + 21 KB staging
static_assert(RB_W + 2 * RB_NL <= 63 && RB_W_EPI + RB_EPI_OPS <= 63 && RB_LA < RB_D, "vmcnt is a 6-bit counter; a slot is re-filled one step AFTER the step that consumed it");

// Register classes.  The accumulators take 160 of the 256 AGPRs; the B ring (200 registers) + slab + A fragments + addresses do not fit the 256 arch
// VGPRs, and a spill (or a copy into a spare AGPR) of a ring register is fatal here: the loads are hidden from the compiler, which would move the
// register's OLD content while the load is still in flight.  So ring slots RB_AG_FROM.. live in AGPRs by constraint -- global_load writes AGPRs directly
// and the MFMA reads its A / B operands from either file on gfx90a+ -- and nothing is left for the allocator to spill.
#ifndef RB_AG_FROM_VALUE
#define RB_AG_FROM_VALUE 5
#endif
constexpr int RB_AG_FROM = RB_AG_FROM_VALUE;
static __device__ __forceinline__ void rb_load16(f16x8& dst, const f16* src) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
}
template <bool AG>
static __device__ __forceinline__ void rb_load16s(f16x8& dst, uint32_t off, const char* base) {
    if constexpr (AG) asm volatile("global_load_dwordx4 %0, %1, %2" : "=a"(dst) : "v"(off), "s"(base) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory");
}
// the wait that makes a ring slot's registers usable: the asm statement "redefines" them, so no consumer can be scheduled above it
template <int N, bool AG>
static __device__ __forceinline__ void rb_wait_slot(f16x8 (&b)[RB_TN]) {
    if constexpr (AG) asm volatile("s_waitcnt vmcnt(%5)" : "+a"(b[0]), "+a"(b[1]), "+a"(b[2]), "+a"(b[3]), "+a"(b[4]) : "n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(%5)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]) : "n"(N) : "memory");
}

#ifndef RBK_ONLY
template <int EPI>   // 0: gemm_epilogue_lds (bias / colscale / row bias / activation / residual), 1: GEGLU
__global__ __launch_bounds__(RB_NW * 64, 1) void gemm_rb_kernel(fd_gemm_desc p, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) f16 smem[];
    const f16* zp = fd_zero_page;
    asm volatile("" : "+s"(zp));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;

    // this workgroup's tile range; tile L = column block L / ntm, row block L % ntm
    const int ntiles = ntm * ntn, G = gridDim.x, w = xcd_remap(blockIdx.x, G);
    const int base = ntiles / G, rem = ntiles - base * G;
    const int Lbeg = w * base + min(w, rem), Lend = Lbeg + base + (w < rem ? 1 : 0);

    const f16* A = (const f16*)p.A;
    const f16* B = (const f16*)p.B;
    const f16* A2 = (const f16*)p.A2;
    const f16* B2 = (const f16*)p.B2;
    // main k-steps: K / 32 (a multiple of ten) and, when the second slab is a full operand (channel concatenation, K2 % 320 == 0), its steps too;
    // a short second slab (the LoRA rank update, K2 <= 32) is the one "slab step" behind them
    const int nk1 = p.K >> 5;
    const bool slab = p.K2 > 0 && p.K2 <= 32;
    const int nkm = nk1 + (slab ? 0 : (p.K2 >> 5));

    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const uint32_t frag = lds0 + (uint32_t)(l15 * 32 + ((lg ^ swz_g(l15 >> 2)) * 8)) * 2;
    constexpr uint32_t STAGE_B = RB_STAGE * 2, GROUP_B = PP_GROUP * 2;
    f16* const slab_lds = smem + RB_NSTA * RB_STAGE;
    const uint32_t slab_frag = frag + RB_NSTA * STAGE_B;
    f16* const slabb_lds = slab_lds + RB_STAGE;                     // [4 waves][5 groups of 16 columns x 64 bytes]
    const uint32_t slabb_frag = slab_frag + STAGE_B + (uint32_t)(wave * RB_TN) * GROUP_B;
    f16* const wave_lds = slabb_lds + RB_NW * RB_TN * PP_GROUP + wave * RB_EPI_HALFS;
    f16* const dump = slabb_lds + RB_NW * RB_TN * PP_GROUP + RB_NW * RB_EPI_HALFS;

    // ---- issue side: runs ten main k-steps ahead of the multiplies, so it crosses into the next tile first.  Addresses are a wave-uniform 64-bit base
    // (operand pointer + k offset: scalar arithmetic) plus a per-lane 32-bit byte offset that only changes with the tile (A rows) or the column block /
    // K segment (B rows): a k-step costs no vector address arithmetic.  Rows beyond M are clamped to row M - 1 (their products are never stored) and
    // steps beyond the workgroup's last tile re-read its last addresses, so the per-step operation count stays uniform without a zero page.
    const int lrow = lane >> 2;
    const int kchunk = ((lane & 3) ^ swz_g(lane >> 4)) * 8;
    int iL = Lbeg, ik = 0, ia = 0;                                  // ia: A ring slot of the next stage to issue
    uint32_t a_off[2] = {0, 0}, b_off[RB_TN] = {0, 0, 0, 0, 0};     // byte offsets of this lane's A rows / B rows in the current segment
    int i_mt = 0, i_nt = 0;
    auto set_offsets = [&] {                              // at a tile change and at a segment change
        const bool seg2 = ik >= nk1;
        const int64_t la = seg2 ? p.lda2 : p.lda, lb = seg2 ? p.ldb2 : p.ldb;
#pragma unroll
        for (int i = 0; i < RB_AG; ++i) {
            const int m = min(i_mt * RB_BM + min(wave + RB_NW * i, RB_TM - 1) * 16 + lrow, p.M - 1);
            a_off[i] = (uint32_t)(((int64_t)m * la + kchunk) * 2);
        }
#pragma unroll
        for (int j = 0; j < RB_TN; ++j) b_off[j] = (uint32_t)(((int64_t)(i_nt * RB_BN + wave * 80 + 16 * j + l15) * lb + lg * 8) * 2);
    };
    auto set_tile = [&](int L) {
        if (L < Lend) {
            i_nt = L / ntm;
            i_mt = L - i_nt * ntm;
        }
        set_offsets();
    };
    set_tile(iL);
    auto seg_base = [&](const f16* P1, const f16* P2) {   // wave-uniform: operand pointer of the current segment advanced to the current k-step
        const bool seg2 = ik >= nk1;
        return (const char*)(seg2 ? P2 : P1) + (int64_t)(seg2 ? ik - nk1 : ik) * 64;
    };
    auto issue_b = [&](f16x8& dst, int j, auto agc) {     // B fragment j of the main k-step at (iL, ik)
        rb_load16s<decltype(agc)::value>(dst, b_off[j], seg_base(B, B2));
    };
    auto issue_a = [&] {                                  // the wave's two A groups of the main k-step at (iL, ik), then on to the next step
        const char* ab = seg_base(A, A2);
        f16* st = smem + ia * RB_STAGE;
#pragma unroll
        for (int i = 0; i < RB_AG; ++i)        // a group index beyond the tile (96-row tiles: waves 2, 3) lands in the dump group: the operation count stays uniform
            glds16((const f16*)(ab + a_off[i]), (wave + RB_NW * i) < RB_TM ? st + (wave + RB_NW * i) * PP_GROUP : dump);
        ia = ia + 1 == RB_NSTA ? 0 : ia + 1;
        if (++ik == nkm) {
            ik = 0;
            ++iL;
            set_tile(iL);
        } else if (ik == nk1) set_offsets();              // into the second (full) K slab: its own leading dimensions
    };

    f32x4 acc[RB_TM][RB_TN];
    f16x8 breg[RB_D][RB_TN], bsl[RB_TN];
    f16x8 areg[RB_TM];

    // the LoRA slab of tile L (one k-step behind the main ones): t[m, K2] (A2) and up[n, K2] (B2) into their own LDS stages -- both as 16-row x 64-byte groups, so
    // the slab step reads its B fragments with the same fragment address as A fragments; k beyond K2 reads the zero page.  2 + 5 operations per wave.
    auto issue_slab = [&](int L) {
        const int nt = L / ntm, mt = L - nt * ntm;
        const bool kok = kchunk < p.K2;
#pragma unroll
        for (int i = 0; i < RB_AG; ++i) {
            const int m = mt * RB_BM + (wave + RB_NW * i) * 16 + lrow;
            glds16((m < p.M && kok) ? A2 + (int64_t)m * p.lda2 + kchunk : zp, (wave + RB_NW * i) < RB_TM ? slab_lds + (wave + RB_NW * i) * PP_GROUP : dump);
        }
#pragma unroll
        for (int j = 0; j < RB_TN; ++j)
            glds16(kok ? B2 + (int64_t)(nt * RB_BN + wave * 80 + 16 * j + lrow) * p.ldb2 + kchunk : zp, slabb_lds + (wave * RB_TN + j) * PP_GROUP);
    };

    // prologue: ten main k-steps in flight, the first one landed and its A fragments read
    static_for<0, RB_LA>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
#pragma unroll
        for (int j = 0; j < RB_TN; ++j) issue_b(breg[s][j], j, std::integral_constant<bool, (s >= RB_AG_FROM)>{});
        issue_a();
    });
    rb_wait_slot<(RB_LA - 1) * RB_NL, false>(breg[0]);
    raw_barrier();
    static_for<0, RB_TM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        ds_read16<i * GROUP_B>(areg[i], frag);
    });
    wait_lgkm<0>();
#pragma unroll
    for (int i = 0; i < RB_TM; ++i) tie(areg[i]);

    int ca = 0;                                           // A ring slot of the stage whose fragments sit in areg
    uint32_t a_cur = frag;                                // LDS byte address (fragment 0, this lane) of that stage
    // One k-step: wait until the NEXT step to be multiplied has landed (vmcnt + barrier), multiply this one (bf = its five B fragments) fragment row by
    // fragment row, reading the next stage (LDS byte address a_next) into areg underneath.  ZERO: the first k-step of a tile starts the accumulators from the
    // constant 0 (no zeroing pass, and the accumulators are not live across tiles).
    //
    // MFMAs execute long after they issue (the wave runs ahead of the matrix pipe by up to a k-step's worth of MFMAs), and nothing interlocks a returning
    // load against a queued MFMA that still has to READ its destination: a global_load into a B slot issued right behind the step that multiplied it
    // clobbered the sources of that step's last MFMAs (products of the NEXT revolution's data in the last fragment row / column, more of them the colder
    // the operands: profiles/r05_gemm_rb_bringup.txt).  Hence the distances kept here: a B slot is re-filled at the END of the step AFTER the one that
    // consumed it; fragment row i of the next stage is read behind the MFMAs of row i + 1, the last row at the top of the next step.
    auto mma_step = [&](auto zero_c, f16x8 (&bf)[RB_TN], f16x8 (&bnext)[RB_TN], auto next_agc, uint32_t a_next, auto&& extra_reads, auto&& refill) {
        constexpr bool ZERO = decltype(zero_c)::value;
        rb_wait_slot<RB_W, decltype(next_agc)::value>(bnext);
        raw_barrier();                                    // every wave's A groups of the next step have landed; every wave has finished the previous step
        ds_read16<(RB_TM - 1) * GROUP_B>(areg[RB_TM - 1], a_cur);      // the last fragment row of THIS stage (its registers served the previous step until now)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        static_for<0, RB_TM>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i == RB_TM - 1) {
                wait_lgkm<0>();
                tie(areg[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < RB_TN; ++j) {
                if constexpr (ZERO) acc[i][j] = FD_MFMA_16x16x32(bf[j], areg[i], ((f32x4){0.f, 0.f, 0.f, 0.f}));
                else acc[i][j] = FD_MFMA_16x16x32(bf[j], areg[i], acc[i][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i >= 1 && i < RB_TM - 1) {
                ds_read16<(i - 1) * GROUP_B>(areg[i - 1], a_next);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        __builtin_amdgcn_s_setprio(0);
        ds_read16<(RB_TM - 2) * GROUP_B>(areg[RB_TM - 2], a_next);     // behind the whole last row: five MFMAs and the re-fill below in between
        extra_reads();
        refill();                                         // main step g + 9 into the slot the PREVIOUS step consumed
        a_cur = a_next;
        wait_lgkm<0>();
#pragma unroll
        for (int i = 0; i < RB_TM - 1; ++i) tie(areg[i]);
        __builtin_amdgcn_sched_barrier(0);
    };
    // ten main k-steps (one ring revolution) starting at main step kb of the tile
    auto revolution = [&](auto first_c, int kb) {
        static_for<0, RB_D>([&](auto rc) {
            constexpr int r = decltype(rc)::value, rn = (r + 1) % RB_D;
            const bool to_slab = slab && r == RB_D - 1 && kb + RB_D == nkm;      // the step after this one is the tile's slab step
            const int na = ca + 1 == RB_NSTA ? 0 : ca + 1;
            const uint32_t a_next = to_slab ? slab_frag : frag + (uint32_t)na * STAGE_B;
            mma_step(std::integral_constant<bool, (decltype(first_c)::value && r == 0)>{}, breg[r], breg[rn], std::integral_constant<bool, (rn >= RB_AG_FROM)>{}, a_next,
                     [&] {
                         if constexpr (r == RB_D - 1) {   // the slab's B fragments (only meaningful when to_slab; the area always exists)
                             static_for<0, RB_TN>([&](auto jc) {
                                 constexpr int j = decltype(jc)::value;
                                 ds_read16<j * GROUP_B>(bsl[j], slabb_frag);
                             });
                         }
                     },
                     [&] {
                         constexpr int rf = (r + RB_LA) % RB_D;
#pragma unroll
                         for (int j = 0; j < RB_TN; ++j) issue_b(breg[rf][j], j, std::integral_constant<bool, (rf >= RB_AG_FROM)>{});
                         issue_a();
                     });
            if (!to_slab) ca = na;
        });
    };

    for (int L = Lbeg; L < Lend; ++L) {
        if (slab) {                                       // after the previous tile's slab step, ahead of this tile's first multiply
            wait_vm<RB_W>();                              // room for its seven operations (the previous epilogue's stores may still be in flight)
            issue_slab(L);
        }
        revolution(std::true_type{}, 0);
        for (int kb = RB_D; kb < nkm; kb += RB_D) revolution(std::false_type{}, kb);
        if (slab) {
            const int na = ca + 1 == RB_NSTA ? 0 : ca + 1;
#pragma unroll
            for (int j = 0; j < RB_TN; ++j) tie(bsl[j]);  // read behind the last main step's MFMAs; that step's lgkmcnt(0) covered them
            mma_step(std::false_type{}, bsl, breg[0], std::false_type{}, frag + (uint32_t)na * STAGE_B, [] {}, [] {});
            ca = na;
        }
        const int nt = L / ntm, mt = L - nt * ntm;
        const int m0 = mt * RB_BM, n0 = nt * RB_BN + wave * 80;
        wait_vm<RB_W_EPI>();                              // room for the epilogue's own operations under the 6-bit counter
        if (EPI == 1) gemm_epilogue_geglu_lds<RB_TM, RB_TN, RB_TMC>(p, acc, wave_lds, m0, n0, lane);
        else gemm_epilogue_lds<RB_TM, RB_TN, RB_TMC, false>(p, acc, wave_lds, m0, n0, lane, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the rings' run-ahead loads must have landed before this workgroup's LDS / registers are handed on
}

#endif  // !RBK_ONLY

// ======================================================================================= K = 320: B RESIDENT in registers
// The streaming ring above is correct but does not pay (profiles/r05_gemm_rb_streaming_ring_isolated_ab.txt: 0.5 - 0.9x of the round-4 kernels): the dense
// kernels are bound by the BYTES a CU pulls through L2 / the Infinity Cache (~25 KB/us per CU whatever is in flight), and a 96-row tile re-fetches its 320 x K
// weight panel 2.7x as often as a 256-row one.  Where the whole panel of a workgroup's column block fits the ring -- K = 320: ten k-steps x five fragments =
// 200 registers per wave, every projection of the 64^2 level -- it is fetched ONCE per workgroup and the kernel only streams A (6 KB per k-step and CU
// instead of 26): proj_in / to_q / to_out / proj_out / FF1 / the K = 320 data gradients.
//   * workgroup <-> (column block c, row range): blockIdx b runs on XCD b % 8 (speed only); the 32 workgroups of an XCD are dealt (c = k % ntn, q = k / ntn)
//     with k = b / 8, so the ntn workgroups that read the SAME A rows for different column blocks sit on one XCD at the same time (one L2 fill, ntn - 1 hits);
//   * per k-step a wave issues its two A groups only, eleven stages ahead (22 operations outstanding: far from the 6-bit vmcnt limit);
//   * everything else -- fragment reads, WAR distances, slab step, epilogues -- is the streaming kernel's.
#ifdef RBK_ONLY
// Epilogue of the resident kernel (bias / colscale, optional residual, fp16 output): gemm_epilogue_lds's plain path with the bias vector read from LDS (staged
// once per workgroup).  A bias fetched from global memory per tile is a load the compiler waits for with vmcnt(0) -- it cannot see the ring -- i.e. a full
// drain of the eleven A stages in flight at every tile.  Returns nothing; issues exactly RBK_NS_PLAIN store instructions for a full tile.
template <int TM, int TN, int TMC>
static __device__ __forceinline__ void rbk_epilogue_plain(const fd_gemm_desc& p, f32x4 (&acc)[TM][TN], f16* wave_lds, const float* bias_lds /* this wave's 80 */,
                                                          int mbase, int nbase, int lane) {
    constexpr int WTN = TN * 16, WTMC = TMC * 16, LDW = WTN + 4, CPR = WTN / 8, RPI = 64 / CPR;
    const int l15 = lane & 15, lg = lane >> 4;
    const int cr = lane / CPR, cc = (lane % CPR) * 8;
    const f16* R = (const f16*)p.residual;
#pragma unroll
    for (int c0 = 0; c0 < TM; c0 += TMC) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const f32x4 bv = *(const f32x4*)(bias_lds + j * 16 + lg * 4);
            const float cs = (nbase + j * 16 + lg * 4) < p.colscale_cols ? p.colscale : 1.f;
#pragma unroll
            for (int ii = 0; ii < TMC; ++ii) {
                const f32x4 v = acc[c0 + ii][j] * cs + bv;
                *(f16x4*)(wave_lds + (ii * 16 + l15) * LDW + j * 16 + lg * 4) = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r0 = 0; r0 < WTMC; r0 += RPI) {
            const int row = r0 + cr;
            const int m = mbase + c0 * 16 + row, n = nbase + cc;
            if (cr < RPI && row < WTMC && m < p.M) {
                const f16x4 lo = *(const f16x4*)(wave_lds + row * LDW + cc);
                const f16x4 hi = *(const f16x4*)(wave_lds + row * LDW + cc + 4);
                f16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (R) {
                    const f16x8 rv = *(const f16x8*)(R + (int64_t)m * p.ldr + n);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (f16)((float)v[k] + (float)rv[k]);
                }
                *(f16x8*)((f16*)p.C + (int64_t)m * p.ldc + n) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}
constexpr int RBK_NS_PLAIN = (RB_TM / RB_TMC) * ((RB_TMC * 16 + 5) / 6);      // 3 passes x 6 store instructions (six rows of ten 16-byte chunks per instruction)
constexpr int RBK_NS_GEGLU = (RB_TM / RB_TMC) * ((RB_TMC * 16 + 11) / 12);    // gated output only (40 columns per wave: twelve rows per instruction)

constexpr int RBK_LA = 10;                                         // A stages issued ahead of the step being multiplied
constexpr int RBK_W = (RBK_LA - 2) * RB_AG;                        // at the top of step g the stages g+2 .. g+LA-1 may stay outstanding

template <int EPI>
__global__ __launch_bounds__(RB_NW * 64, 1) void gemm_rbk_kernel(fd_gemm_desc p, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) f16 smem[];
    const f16* zp = fd_zero_page;
    asm volatile("" : "+s"(zp));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;

    const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;           // 32 workgroups per XCD
    const int gq = 32 / ntn;                                        // row-range groups per XCD
    const int nt = kx % ntn, q = kx / ntn;
    if (q >= gq) return;                                            // 32 % ntn workgroups per XCD have no work (uniform: before any barrier)
    const int R = 8 * gq, rr = xcd * gq + q;
    const int Tbeg = (int)((int64_t)ntm * rr / R), Tend = (int)((int64_t)ntm * (rr + 1) / R);
    if (Tbeg >= Tend) return;

    const f16* A = (const f16*)p.A;
    const f16* B = (const f16*)p.B;
    const f16* A2 = (const f16*)p.A2;
    const f16* B2 = (const f16*)p.B2;
    const bool slab = p.K2 > 0;

    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const uint32_t frag = lds0 + (uint32_t)(l15 * 32 + ((lg ^ swz_g(l15 >> 2)) * 8)) * 2;
    constexpr uint32_t STAGE_B = RB_STAGE * 2, GROUP_B = PP_GROUP * 2;
    f16* const slab_lds = smem + RB_NSTA * RB_STAGE;
    const uint32_t slab_frag = frag + RB_NSTA * STAGE_B;
    f16* const slabb_lds = slab_lds + RB_STAGE;
    const uint32_t slabb_frag = slab_frag + STAGE_B + (uint32_t)(wave * RB_TN) * GROUP_B;
    f16* const wave_lds = slabb_lds + RB_NW * RB_TN * PP_GROUP + wave * RB_EPI_HALFS;
    f16* const dump = slabb_lds + RB_NW * RB_TN * PP_GROUP + RB_NW * RB_EPI_HALFS;

    // ---- issue side (A only): row block it, k-step ik, ring slot ia
    const int lrow = lane >> 2;
    const int kchunk = ((lane & 3) ^ swz_g(lane >> 4)) * 8;
    int it = Tbeg, ik = 0, ia = 0;
    uint32_t a_off[RB_AG];
    auto set_rows = [&] {
        const int t = min(it, Tend - 1);                  // beyond the range: re-read the last tile (uniform operation count, nothing of it is multiplied)
#pragma unroll
        for (int i = 0; i < RB_AG; ++i) {
            const int m = min(t * RB_BM + min(wave + RB_NW * i, RB_TM - 1) * 16 + lrow, p.M - 1);
            a_off[i] = (uint32_t)(((int64_t)m * p.lda + kchunk) * 2);
        }
    };
    set_rows();
    auto issue_a = [&] {
        const char* ab = (const char*)A + (int64_t)ik * 64;
        f16* st = smem + ia * RB_STAGE;
#pragma unroll
        for (int i = 0; i < RB_AG; ++i) glds16((const f16*)(ab + a_off[i]), (wave + RB_NW * i) < RB_TM ? st + (wave + RB_NW * i) * PP_GROUP : dump);
        ia = ia + 1 == RB_NSTA ? 0 : ia + 1;
        if (++ik == RB_D) {
            ik = 0;
            ++it;
            set_rows();
        }
    };
    auto issue_slab = [&](int t) {
        const bool kok = kchunk < p.K2;
#pragma unroll
        for (int i = 0; i < RB_AG; ++i) {
            const int m = t * RB_BM + (wave + RB_NW * i) * 16 + lrow;
            glds16((m < p.M && kok) ? A2 + (int64_t)m * p.lda2 + kchunk : zp, (wave + RB_NW * i) < RB_TM ? slab_lds + (wave + RB_NW * i) * PP_GROUP : dump);
        }
#pragma unroll
        for (int j = 0; j < RB_TN; ++j)
            glds16(kok ? B2 + (int64_t)(nt * RB_BN + wave * 80 + 16 * j + lrow) * p.ldb2 + kchunk : zp, slabb_lds + (wave * RB_TN + j) * PP_GROUP);
    };

    f32x4 acc[RB_TM][RB_TN];
    f16x8 breg[RB_D][RB_TN], bsl[RB_TN];
    f16x8 areg[RB_TM];

    // prologue: the column block's whole weight panel (ten k-steps x five fragments, 50 operations), then the first A stages behind it
    {
        const uint32_t boff = (uint32_t)(((int64_t)(nt * RB_BN + wave * 80 + l15) * p.ldb + lg * 8) * 2);
        static_for<0, RB_D>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
#pragma unroll
            for (int j = 0; j < RB_TN; ++j)
                rb_load16s<(s >= RB_AG_FROM)>(breg[s][j], boff + (uint32_t)(16 * j * p.ldb * 2), (const char*)B + s * 64);
        });
    }
    for (int s = 0; s < 6; ++s) issue_a();                // 50 + 12 operations outstanding
    static_for<0, RB_D>([&](auto sc) {                    // every B fragment has landed (they are older than the twelve A groups)
        constexpr int s = decltype(sc)::value;
        rb_wait_slot<6 * RB_AG, (s >= RB_AG_FROM)>(breg[s]);
    });
    for (int s = 6; s < RBK_LA; ++s) issue_a();
    wait_vm<(RBK_LA - 1) * RB_AG>();                      // stage 0 landed
    raw_barrier();
    static_for<0, RB_TM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        ds_read16<i * GROUP_B>(areg[i], frag);
    });
    wait_lgkm<0>();
#pragma unroll
    for (int i = 0; i < RB_TM; ++i) tie(areg[i]);

    int ca = 0;
    uint32_t a_cur = frag;
    // Operations that may stay outstanding at the top of a step = everything issued after the A groups of the stage that must have landed.  vmcnt retires
    // in issue order, STORES INCLUDED: counted strictly (16: the eight younger stages), every step behind an epilogue waited for two more of its 18 stores --
    // a tile paced by store latency (first form of this kernel: ~1 us per k-step).  Behind a FULL tile's epilogue the younger operations are exactly
    // 16 + the epilogue's store instructions (+ 7 with a slab bundle): a constant per launch, selected per tile (the first tile of a workgroup and the tile
    // behind a ragged one count strictly -- too small a number only waits longer, too large a number reads stale data).
    auto mma_step = [&](auto zero_c, f16x8 (&bf)[RB_TN], uint32_t a_next, auto&& extra_reads, bool refill, int loose) {
        constexpr bool ZERO = decltype(zero_c)::value;
        if (loose == 2) wait_vm<RBK_W + RB_NL + (EPI == 1 ? RBK_NS_GEGLU : RBK_NS_PLAIN)>();
        else if (loose == 1) wait_vm<RBK_W + (EPI == 1 ? RBK_NS_GEGLU : RBK_NS_PLAIN)>();
        else wait_vm<RBK_W>();
        raw_barrier();                                    // every wave's A groups of the next step have landed; every wave has finished the previous step
        ds_read16<(RB_TM - 1) * GROUP_B>(areg[RB_TM - 1], a_cur);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        static_for<0, RB_TM>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i == RB_TM - 1) {
                wait_lgkm<0>();
                tie(areg[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < RB_TN; ++j) {
                if constexpr (ZERO) acc[i][j] = FD_MFMA_16x16x32(bf[j], areg[i], ((f32x4){0.f, 0.f, 0.f, 0.f}));
                else acc[i][j] = FD_MFMA_16x16x32(bf[j], areg[i], acc[i][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i >= 1 && i < RB_TM - 1) {
                ds_read16<(i - 1) * GROUP_B>(areg[i - 1], a_next);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        __builtin_amdgcn_s_setprio(0);
        ds_read16<(RB_TM - 2) * GROUP_B>(areg[RB_TM - 2], a_next);
        extra_reads();
        if (refill) issue_a();
        a_cur = a_next;
        wait_lgkm<0>();
#pragma unroll
        for (int i = 0; i < RB_TM - 1; ++i) tie(areg[i]);
        __builtin_amdgcn_sched_barrier(0);
    };

    // exact store counts exist for the epilogues written for this kernel: bias (+ colscale, + residual) and the gated output without the recorded pre-gate
    // projection; everything else (activations, row bias, recorded GEGLU) goes through gemm_device.h's epilogues and counts strictly
    const bool own_epi = EPI == 1 ? p.residual == nullptr : (p.act == FD_ACT_NONE && !p.rowbias && p.alpha == 1.f);
    float* const bias_lds = (float*)(dump + PP_GROUP);    // [320]: this column block's bias (zeros without one), staged once
    for (int i = tid; i < RB_BN; i += RB_NW * 64) bias_lds[i] = p.bias ? p.bias[nt * RB_BN + i] : 0.f;
    __syncthreads();
    int loose = 0;
    for (int t = Tbeg; t < Tend; ++t) {
        if (slab) issue_slab(t);                          // 7 operations; with the ring's 20 and an epilogue's 34 still below the 6-bit limit
        const int lw = loose ? (slab ? 2 : 1) : 0;
        static_for<0, RB_D>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const bool to_slab = slab && r == RB_D - 1;
            const int na = ca + 1 == RB_NSTA ? 0 : ca + 1;
            const uint32_t a_next = to_slab ? slab_frag : frag + (uint32_t)na * STAGE_B;
            mma_step(std::integral_constant<bool, (r == 0)>{}, breg[r], a_next,
                     [&] {
                         if constexpr (r == RB_D - 1) {
                             static_for<0, RB_TN>([&](auto jc) {
                                 constexpr int j = decltype(jc)::value;
                                 ds_read16<j * GROUP_B>(bsl[j], slabb_frag);
                             });
                         }
                     },
                     true, r == RB_D - 1 ? 0 : lw);      // the stage step 9 waits for was issued behind the previous epilogue: strict count
            if (!to_slab) ca = na;
        });
        if (slab) {
            const int na = ca + 1 == RB_NSTA ? 0 : ca + 1;
#pragma unroll
            for (int j = 0; j < RB_TN; ++j) tie(bsl[j]);
            mma_step(std::false_type{}, bsl, frag + (uint32_t)na * STAGE_B, [] {}, false, 0);
            ca = na;
        }
        const int m0 = t * RB_BM, n0 = nt * RB_BN + wave * 80;
        if (EPI == 1) gemm_epilogue_geglu_lds<RB_TM, RB_TN, RB_TMC>(p, acc, wave_lds, m0, n0, lane);
        else if (own_epi) rbk_epilogue_plain<RB_TM, RB_TN, RB_TMC>(p, acc, wave_lds, bias_lds + wave * 80, m0, n0, lane);
        else gemm_epilogue_lds<RB_TM, RB_TN, RB_TMC, false>(p, acc, wave_lds, m0, n0, lane, 0, 0);
        loose = 0;   // MEASURED UNSAFE (376 wrong outputs on 28840 x 320 x 320): stores retire out of order with older loads, so only the strict count holds
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#endif  // RBK_ONLY

#ifndef RBK_ONLY
// Shapes the register-B kernel takes: unbatched dense fp16 GEMMs with K % 320 == 0 (ten k-steps per ring revolution), N % 320 == 0, an optional second slab
// that is either a LoRA rank update (K2 <= 32) or a full operand (K2 % 320 == 0), and an output that can go through the LDS-staged epilogues.  GroupNorm
// statistics / LayerNorm second outputs, convolutions, fp32 outputs and split-K stay on the other kernels.
bool fd_gemm_rb_eligible(const fd_gemm_desc& d) {
    if (d.conv || d.batch > 1 || (d.N % RB_BN) != 0 || d.K <= 0 || (d.K % 320) != 0) return false;
    if (d.K2 > 0 && !((d.K2 <= 32 && (d.K2 & 7) == 0) || (d.K2 % 320) == 0)) return false;
    if (d.out_dtype != FD_OUT_F16 || (d.ldc & 7) != 0 || (d.residual && (d.ldr & 7) != 0) || (d.rowbias && (d.ld_rowbias & 3) != 0)) return false;
    if (d.gn_stats || d.ln_out) return false;
    return true;
}

int fd_gemm_launch_rb(const fd_gemm_desc& d, hipStream_t s, int max_wg) {
    const int ntm = (d.M + RB_BM - 1) / RB_BM, ntn = d.N / RB_BN;
    const int ntiles = ntm * ntn, G = ntiles < max_wg ? ntiles : max_wg;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)gemm_rb_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rb_lds());
        (void)hipFuncSetAttribute((const void*)gemm_rb_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rb_lds());
    });
    if (d.act == FD_ACT_GEGLU) hipLaunchKernelGGL((gemm_rb_kernel<1>), dim3(G), dim3(RB_NW * 64), rb_lds(), s, d, ntm, ntn);
    else hipLaunchKernelGGL((gemm_rb_kernel<0>), dim3(G), dim3(RB_NW * 64), rb_lds(), s, d, ntm, ntn);
    return fd_check_launch("fd_gemm(rb)");
}
#endif  // !RBK_ONLY

#ifdef RBK_ONLY
bool fd_gemm_rb_eligible(const fd_gemm_desc& d);
// the resident form: the streaming form's shapes with K == 320, a LoRA slab at most, and no more column blocks than an XCD has workgroups
bool fd_gemm_rbk_eligible(const fd_gemm_desc& d) { return fd_gemm_rb_eligible(d) && d.K == 320 && d.K2 <= 32 && d.N / RB_BN <= 32; }

int fd_gemm_launch_rbk(const fd_gemm_desc& d, hipStream_t s) {
    const int ntm = (d.M + RB_BM - 1) / RB_BM, ntn = d.N / RB_BN;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)gemm_rbk_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rb_lds());
        (void)hipFuncSetAttribute((const void*)gemm_rbk_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rb_lds());
    });
    if (d.act == FD_ACT_GEGLU) hipLaunchKernelGGL((gemm_rbk_kernel<1>), dim3(256), dim3(RB_NW * 64), rb_lds(), s, d, ntm, ntn);
    else hipLaunchKernelGGL((gemm_rbk_kernel<0>), dim3(256), dim3(RB_NW * 64), rb_lds(), s, d, ntm, ntn);
    return fd_check_launch("fd_gemm(rbk)");
}
#endif  // RBK_ONLY
