// Device pieces shared by the ping-pong GEMM kernels (gemm_pp.hip: one tile per workgroup; gemm_pps.hip: persistent workgroups that stream
// tiles): counted vmcnt / raw barrier helpers, the pinned k = 32 fragment-streaming step with a mid-step hook, the staging geometry.
#pragma once
#include "gemm_device.h"

template <int N>
static __device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
static __device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

// one k = 32 step of a wave tile: the operand with fewer fragments resident, the other streamed PD ahead (mma_k32 of gemm_device.h), with
// a hook that runs between the two halves of the step -- the point where the leading group crosses the workgroup barrier.
// PRIO: s_setprio(1) from the first to the last MFMA of the step -- with the two waves of a SIMD in different roles the arbiter has something
// to decide (the wave in its MFMA stream outranks the one doing boundary work); in a lockstep loop it is a no-op
// ABL (measurement builds only, FD_GEMM_DBG): 0 = the kernel, 2 = no MFMAs (fragment reads, waits, barriers and DMA issue only), 6 = no fragment reads (MFMAs on
// whatever the registers hold)
// byte offset of 16-row fragment group i behind the lane's base address: i * GS for the 16-row-group staging image; AW > 0 (gemm_halo.hip): the A operand
// is a halo image of W = AW pixels per image row, 64 bytes per pixel, AW + 8 pixel slots per LDS row -- fragment i starts 16 i pixels further on, plus
// 8 slots for every image row crossed
template <int AW, int GS>
__host__ __device__ constexpr int pp_frag_off(int i) { return AW ? (16 * i + 8 * ((16 * i) / (AW ? AW : 1))) * 64 : i * GS; }

template <int TM, int TN, int PD, int GS, bool PRIO, class MID, int ABL = 0, int AW = 0>
static __device__ __forceinline__ void mma_k32_mid(f32x4 (&acc)[TM][TN], uint32_t a_addr, uint32_t b_addr, MID&& mid) {
    constexpr bool BRES = TN <= TM;   // the operand with fewer fragments stays resident for the step
    constexpr int NR = BRES ? TN : TM, NS = BRES ? TM : TN, R = PD + 1;
    constexpr int RW = BRES ? 0 : AW, SW = BRES ? AW : 0;     // which of the two is A
    const uint32_t r_addr = BRES ? b_addr : a_addr, s_addr = BRES ? a_addr : b_addr;
    f16x8 res[NR], ring[R];
    if constexpr (ABL == 6) {
#pragma unroll
        for (int i = 0; i < NR; ++i) asm volatile("" : "=v"(res[i]));
#pragma unroll
        for (int i = 0; i < R; ++i) asm volatile("" : "=v"(ring[i]));
    }
    static_for<0, NR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (ABL != 6) ds_read16<pp_frag_off<RW, GS>(i)>(res[i], r_addr);
    });
    static_for<0, PD>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (ABL != 6) ds_read16<pp_frag_off<SW, GS>(i)>(ring[i % R], s_addr);
    });
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    static_for<0, NS>([&](auto ic) {
        constexpr int s = decltype(ic)::value;
        if constexpr (s + PD < NS && ABL != 6) ds_read16<pp_frag_off<SW, GS>(s + PD)>(ring[(s + PD) % R], s_addr);
        constexpr int after = (NS - 1 - s) < PD ? (NS - 1 - s) : PD;
        wait_lgkm<after>();
        if constexpr (s == 0) {
#pragma unroll
            for (int r = 0; r < NR; ++r) tie(res[r]);
        }
        tie(ring[s % R]);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if constexpr (ABL == 2) asm volatile("" ::"v"(res[r]), "v"(ring[s % R]));       // keep the reads alive
            else if constexpr (BRES) acc[s][r] = FD_MFMA_16x16x32(res[r], ring[s % R], acc[s][r]);
            else acc[r][s] = FD_MFMA_16x16x32(ring[s % R], res[r], acc[r][s]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (s == NS / 2 - 1) {
            mid();
            __builtin_amdgcn_sched_barrier(0);
        }
    });
    if (PRIO) __builtin_amdgcn_s_setprio(0);
}

constexpr int PP_BN = 320, PP_NW = 8, PP_NST = 4;
constexpr int PP_GROUP = 16 * 32;                       // halfs per 16-row group (16 rows x 64 bytes = 1 KB = one global_load_lds_dwordx4)
constexpr int PP_NGB = PP_BN / 16;                      // 20 B groups per k-step
template <int BM> constexpr int pp_stage() { return (BM / 16 + PP_NGB) * PP_GROUP; }      // 36 KB (BM = 256) / 28 KB (BM = 128)
template <int BM> constexpr size_t pp_lds() { return (size_t)(PP_NST * pp_stage<BM>() + PP_NW * PP_GROUP) * sizeof(f16); }   // ring + a dump group per wave

