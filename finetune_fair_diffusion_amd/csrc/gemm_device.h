// Device-side building blocks shared by the GEMM translation units (gemm.hip: 4-wave BK=32 and 8/16-wave BK=64 kernels; gemm_mb.hip:
// the multi-block-per-CU BK=32 kernels): measurement hooks, the zero page, direct-to-LDS loads, the three epilogues, the pinned
// fragment schedule.  Everything here is static / inline: each translation unit gets its own copy (no relocatable device code).
#pragma once
#include "common.h"
#include <stdlib.h>
#include <stdio.h>
#include <mutex>

// Measurement hooks (FD_GEMM_DBG sentinels inside the kernels, tile-policy A/B switches read from the environment) exist only in
// builds made with -DFD_BENCH_HOOKS (``make BENCH_HOOKS=1``): in the product library a stray environment variable can neither
// change tile selection nor skip work.
#ifdef FD_BENCH_HOOKS
#define FD_DBG_IS(p, v) ((p).batch == -(v))
#define FD_DBG_GE(p, v) ((p).batch <= -(v))
static inline const char* bench_env(const char* name) { return getenv(name); }
#else
#define FD_DBG_IS(p, v) false
#define FD_DBG_GE(p, v) false
static inline const char* bench_env(const char*) { return nullptr; }
#endif

struct ConvRow {
    int b, oy, ox;
    bool valid;
};

// ======================================================================================= 4-wave, BK = 32
// Both operand tiles travel global -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write).  The LDS image of a 16-row x 64-byte group is
// exactly what one wave-instruction writes (lane l -> row l>>2, 16-byte slot l&3), so rows are unpadded;
// bank conflicts on the ds_read_b128 fragment reads are removed by permuting WHICH 16-byte k-chunk a slot
// holds (chunk = slot ^ G[(row>>2)&3], G = {0,3,2,1}) on the source address, and reading with the same
// involution.  Out-of-range rows / k-chunks / conv padding read from a zero page instead of branching.
// Two LDS stages: the loads of k-tile t+1 are in flight under the MFMAs of tile t; one barrier per k-tile.
static __device__ __attribute__((aligned(16))) f16 fd_zero_page[64];

__device__ __forceinline__ int swz_g(int r4) { return (4 - r4) & 3; }  // {0,3,2,1}

__device__ __forceinline__ void glds16(const f16* src, f16* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}


// FD_CONV_UP2PI: row of the [Bn, 2H, 2W, N] result that low-res row m = (b, y, x) of phase ``up_phase`` = py * 2 + px lands on (up_phase < 0: m itself)
__device__ __forceinline__ int64_t up2p_row(const fd_gemm_desc& p, int m, int up_phase) {
    if (up_phase < 0) return m;
    const int hw = p.H * p.W;
    const int b = m / hw, r = m - b * hw;
    const int y = r / p.W, x = r - y * p.W;
    return ((int64_t)(b * 2 * p.H + 2 * y + (up_phase >> 1))) * (2 * p.W) + 2 * x + (up_phase & 1);
}

// ---- shared epilogue: lane holds C[m][n..n+3] per 16x16 tile (swapped-operand MFMA layout).  Bias / row-bias / residual
// are fetched as one vector per tile (the epilogue of a short-K GEMM is otherwise more VMEM instructions than its main loop).
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const fd_gemm_desc& p, f32x4 (&acc)[TM][TN], int mbase, int nbase, int l15, int lg,
                                              int64_t zC, int64_t zR, int up_phase = -1) {
    const f16* R = p.residual ? (const f16*)p.residual + zR : nullptr;
    const f16* RB = (const f16*)p.rowbias;
    const bool vec_ok = ((p.N & 3) == 0) && ((p.ldc & 3) == 0) && (!R || (p.ldr & 3) == 0) && (!RB || (p.ld_rowbias & 3) == 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = nbase + j * 16 + lg * 4;
        if (n >= p.N) continue;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            if (vec_ok) bv = *(const f32x4*)(p.bias + n);
            else
                for (int r = 0; r < 4 && n + r < p.N; ++r) bv[r] = p.bias[n + r];
        }
        const float al = n < p.colscale_cols ? p.alpha * p.colscale : p.alpha;     // fd_gemm_desc.colscale (colscale_cols % 4 == 0: uniform over a lane's 4 columns)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = mbase + i * 16 + l15;
            if (m >= p.M) continue;
            float v[4];
            f16x4 rbv = {0, 0, 0, 0}, resv = {0, 0, 0, 0};
            if (vec_ok) {
                if (RB) rbv = *(const f16x4*)(RB + (int64_t)(m / p.rows_per_batch) * p.ld_rowbias + n);
                if (R) resv = *(const f16x4*)(R + (int64_t)m * p.ldr + n);
            } else {
                for (int r = 0; r < 4 && n + r < p.N; ++r) {
                    if (RB) rbv[r] = RB[(int64_t)(m / p.rows_per_batch) * p.ld_rowbias + n + r];
                    if (R) resv[r] = R[(int64_t)m * p.ldr + n + r];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[i][j][r] * al + bv[r] + (float)rbv[r];
                x = apply_act(x, p.act);
                v[r] = x + (float)resv[r];
            }
            const int64_t crow = up2p_row(p, m, up_phase);
            if (p.out_dtype == FD_OUT_F32) {
                float* C = (float*)p.C + zC + crow * p.ldc + n;
                if (vec_ok) *(f32x4*)C = (f32x4){v[0], v[1], v[2], v[3]};
                else
                    for (int r = 0; r < 4 && n + r < p.N; ++r) C[r] = v[r];
            } else {
                f16* C = (f16*)p.C + zC + crow * p.ldc + n;
                if (vec_ok) *(f16x4*)C = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                else
                    for (int r = 0; r < 4 && n + r < p.N; ++r) C[r] = (f16)v[r];
            }
        }
    }
}


// ---- LDS-staged epilogue (fp16 output): the accumulator layout gives each lane 4 consecutive N (8 bytes), i.e. 32-byte
// row segments per store instruction; short-K GEMMs are bound by exactly that store path.  Here each wave parks its
// WTM x WTN tile in LDS (bias / row-bias / activation already applied) and re-reads it as 16 bytes per lane so that a
// store instruction covers whole 128-byte row segments; the residual is added on the way out with 16-byte loads.
template <int TM, int TN, int TMC = TM, bool WSTATS = false>
__device__ __forceinline__ void gemm_epilogue_lds(const fd_gemm_desc& p, f32x4 (&acc)[TM][TN], f16* wave_lds, int mbase, int nbase,
                                                  int lane, int64_t zC, int64_t zR, int up_phase = -1) {
    // TMC: 16-row groups staged per pass (the wave-private LDS region holds TMC*16 rows; big tiles need two passes)
    constexpr int WTN = TN * 16, WTMC = TMC * 16, LDW = WTN + 4;   // +4 halfs: 8-byte aligned rows, spreads the ds_write_b64 banks
    static_assert(TM % TMC == 0, "chunking");
    const int l15 = lane & 15, lg = lane >> 4;
    const f16* RB = (const f16*)p.rowbias;
    const f16* R = p.residual ? (const f16*)p.residual + zR : nullptr;
    constexpr int CPR = WTN / 8;                 // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;                // rows per store instruction
    const int cr = lane / CPR, cc = (lane % CPR) * 8;
    // the epilogue mode is uniform over the launch: branch once, outside the per-element loops (a per-element runtime switch on
    // p.act plus the always-on alpha / row-bias arithmetic made short-K GEMMs VALU-bound here: 149 -> 100 us of epilogue on FF1)
    const bool plain = (p.act == FD_ACT_NONE) && !RB && p.alpha == 1.f && !FD_DBG_IS(p, 5);
    const bool cscale = p.colscale_cols > 0;          // fd_gemm_desc.colscale: its own copy of the plain loop (uniform branch), nothing for the others
    // GroupNorm statistics of the stored tile (fd_gemm_desc.gn_stats), in CANONICAL chunks of 32 rows: whatever kernel and tile produced C, the
    // sums of rows [32 c, 32 c + 32) x 10-channel unit are formed by the same procedure -- lane (cr, column chunk) accumulates the four column PAIRS
    // of its 8 columns over the rows cr, cr + 6, ... of the chunk (v_dot2c: two 16-bit values per fp32 accumulate, no conversions -- three VALU
    // operations per element made the 16-wave short-K kernels 18 % slower, 29 -> 35 us at 65536x320x320), then pair totals over the six row lanes,
    // then lane u < 8 adds the five pairs of unit u -- so a sample's statistics do not depend on the batch it is computed in nor on the tile
    // policy (the CFG-pair prefix evaluates N samples where the duplicated batch has 2N).
    // (WSTATS kernels are separate instantiations -- template code 3 / 4 of gemm_big_kernel, 2 / 3 of gemm_pp_kernel: compiled into the plain ones the
    // extra live registers pushed the 16-wave 256 x 320 kernel, capped at 128, into scratch)
    static_assert(!WSTATS || (WTN == 80 && WTMC % 32 == 0 && 32 * LDW * 2 >= (64 * 8 + WTN) * 4), "statistics epilogue: 80-column wave tiles, passes of whole 32-row chunks");
    constexpr int SUB = WSTATS ? WTMC / 32 : 1;
    float* const gst = WSTATS ? p.gn_stats : nullptr;
    float s0[SUB][4], s1[SUB][4];
    const f16x2 ones = {(f16)1.f, (f16)1.f};
#pragma unroll
    for (int c0 = 0; c0 < TM; c0 += TMC) {
        if (plain && !cscale) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = nbase + j * 16 + lg * 4;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias && n < p.N) bv = *(const f32x4*)(p.bias + n);
#pragma unroll
                for (int ii = 0; ii < TMC; ++ii) {
                    const f32x4 v = acc[c0 + ii][j] + bv;
                    *(f16x4*)(wave_lds + (ii * 16 + l15) * LDW + j * 16 + lg * 4) = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                }
            }
        } else if (plain) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = nbase + j * 16 + lg * 4;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias && n < p.N) bv = *(const f32x4*)(p.bias + n);
                const float cs = n < p.colscale_cols ? p.colscale : 1.f;
#pragma unroll
                for (int ii = 0; ii < TMC; ++ii) {
                    const f32x4 v = acc[c0 + ii][j] * cs + bv;
                    *(f16x4*)(wave_lds + (ii * 16 + l15) * LDW + j * 16 + lg * 4) = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = nbase + j * 16 + lg * 4;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias && n < p.N) bv = *(const f32x4*)(p.bias + n);
                const float al = n < p.colscale_cols ? p.alpha * p.colscale : p.alpha;
#pragma unroll
                for (int ii = 0; ii < TMC; ++ii) {
                    const int i = c0 + ii;
                    const int m = mbase + i * 16 + l15;
                    f16x4 rbv = {0, 0, 0, 0};
                    if (RB && m < p.M && n < p.N) rbv = *(const f16x4*)(RB + (int64_t)(m / p.rows_per_batch) * p.ld_rowbias + n);
                    f16x4 o;
                    if (p.act == FD_ACT_NONE) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (f16)(acc[i][j][r] * al + bv[r] + (float)rbv[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (f16)apply_act(acc[i][j][r] * al + bv[r] + (float)rbv[r], p.act);
                    }
                    if (FD_DBG_IS(p, 5)) o = (f16x4){(f16)acc[i][j][0], (f16)acc[i][j][1], (f16)acc[i][j][2], (f16)acc[i][j][3]};   // FD_GEMM_DBG=5
                    *(f16x4*)(wave_lds + (ii * 16 + l15) * LDW + j * 16 + lg * 4) = o;
                }
            }
        }
        // wave-private region: no block barrier needed, only this wave's own LDS writes must have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (!WSTATS) {
#pragma unroll
            for (int r0 = 0; r0 < WTMC; r0 += RPI) {
                const int row = r0 + cr;
                const int m = mbase + c0 * 16 + row, n = nbase + cc;
                if (cr < RPI && row < WTMC && m < p.M && n < p.N) {
                    // LDS rows are 8-byte aligned (LDW*2 bytes is a multiple of 8): two 8-byte reads
                    const f16x4 lo = *(const f16x4*)(wave_lds + row * LDW + cc);
                    const f16x4 hi = *(const f16x4*)(wave_lds + row * LDW + cc + 4);
                    f16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if (R) {
                        const f16x8 rv = *(const f16x8*)(R + (int64_t)m * p.ldr + n);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = (f16)((float)v[k] + (float)rv[k]);
                    }
                    *(f16x8*)((f16*)p.C + zC + up2p_row(p, m, up_phase) * p.ldc + n) = v;
                }
            }
        } else {
#pragma unroll
            for (int sub = 0; sub < SUB; ++sub) {
#pragma unroll
                for (int k = 0; k < 4; ++k) s0[sub][k] = s1[sub][k] = 0.f;
#pragma unroll
                for (int r0 = 0; r0 < 32; r0 += RPI) {
                    const int row = sub * 32 + r0 + cr;
                    const int m = mbase + c0 * 16 + row, n = nbase + cc;
                    if (cr < RPI && r0 + cr < 32 && m < p.M && n < p.N) {
                        const f16x4 lo = *(const f16x4*)(wave_lds + row * LDW + cc);
                        const f16x4 hi = *(const f16x4*)(wave_lds + row * LDW + cc + 4);
                        f16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        if (R) {
                            const f16x8 rv = *(const f16x8*)(R + (int64_t)m * p.ldr + n);
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[k] = (f16)((float)v[k] + (float)rv[k]);
                        }
                        *(f16x8*)((f16*)p.C + zC + up2p_row(p, m, up_phase) * p.ldc + n) = v;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const f16x2 pr = {v[2 * k], v[2 * k + 1]};
                            s0[sub][k] = FD_DOT2(pr, ones, s0[sub][k]);
                            s1[sub][k] = FD_DOT2(pr, pr, s1[sub][k]);
                        }
                    }
                }
            }
        }
        if (c0 + TMC < TM || gst) {   // the next pass (or the statistics) overwrites the staging rows: this wave's reads must have returned
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (WSTATS && gst) {
            float* red = (float*)wave_lds;          // 64 lanes x 8 floats (+ 40 pair totals x 2): inside the first 32 staging rows
#pragma unroll
            for (int sub = 0; sub < SUB; ++sub) {
                if (cr < RPI) {
                    *(f32x4*)(red + lane * 8) = (f32x4){s0[sub][0], s0[sub][1], s0[sub][2], s0[sub][3]};
                    *(f32x4*)(red + lane * 8 + 4) = (f32x4){s1[sub][0], s1[sub][1], s1[sub][2], s1[sub][3]};
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                // stage 1, lanes 0..39: totals of column pair ``lane`` over the RPI row lanes, ascending; stage 2, lane u < 8: the five pairs of unit u
                float* pairsum = red + 64 * 8;          // [40][2]
                if (lane < WTN / 2) {
                    const int ch = lane >> 2, j = lane & 3;
                    float t0 = 0.f, t1 = 0.f;
#pragma unroll
                    for (int r = 0; r < RPI; ++r) {
                        t0 += red[(r * CPR + ch) * 8 + j];
                        t1 += red[(r * CPR + ch) * 8 + 4 + j];
                    }
                    *(float2*)(pairsum + lane * 2) = make_float2(t0, t1);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const int mrow = mbase + c0 * 16 + sub * 32;
                if (lane < WTN / 10 && mrow < p.M && nbase < p.N) {
                    float t0 = 0.f, t1 = 0.f;
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
                        const float2 v = *(const float2*)(pairsum + (lane * 5 + c) * 2);
                        t0 += v.x;
                        t1 += v.y;
                    }
                    // FD_CONV_UP2PI: the chunks of phase ``up_phase`` follow those of the phases before it (M % 32 == 0)
                    const int64_t slot = (int64_t)((mrow >> 5) + (up_phase > 0 ? up_phase * (p.M >> 5) : 0)) * (p.N / 10) + nbase / 10 + lane;
                    *(float2*)(gst + slot * 2) = make_float2(t0, t1);
                }
                // the next chunk's sums / the next pass's staging overwrite ``red``: the unit lanes' reads must have returned
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

// GEGLU fused into the FF1 projection (no-record forwards): B rows are interleaved (value_c, gate_c), so a lane's four consecutive
// accumulator columns are (v0, g0, v1, g1).  Both halves are rounded to fp16 first, exactly like the unfused projection followed by
// fd_geglu_fwd, so the fused and the unfused path give bit-identical results.  Output tile width is half the GEMM tile width.
template <int TM, int TN, int TMC>
__device__ __forceinline__ void gemm_epilogue_geglu_lds(const fd_gemm_desc& p, f32x4 (&acc)[TM][TN], f16* wave_lds, int mbase, int nbase, int lane) {
    constexpr int WTO = TN * 8, WTMC = TMC * 16, LDW = WTO + 4;
    constexpr int CPR = WTO / 8, RPI = 64 / CPR;
    const int l15 = lane & 15, lg = lane >> 4;
    const int cr = lane / CPR, cc = (lane % CPR) * 8;
    const int No = p.N >> 1;
    f16* AUX = (f16*)p.residual;     // recording forwards: the pre-gate projection (interleaved columns) is kept for the backward
#pragma unroll
    for (int c0 = 0; c0 < TM; c0 += TMC) {
        if (AUX) {
            constexpr int LDWP = TN * 16 + 4, CPRP = TN * 2, RPIP = 64 / CPRP;
            const int crp = lane / CPRP, ccp = (lane % CPRP) * 8;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = nbase + j * 16 + lg * 4;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias && n < p.N) bv = *(const f32x4*)(p.bias + n);
#pragma unroll
                for (int ii = 0; ii < TMC; ++ii) {
                    const f32x4 v = acc[c0 + ii][j] + bv;
                    *(f16x4*)(wave_lds + (ii * 16 + l15) * LDWP + j * 16 + lg * 4) = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r0 = 0; r0 < WTMC; r0 += RPIP) {
                const int row = r0 + crp;
                const int m = mbase + c0 * 16 + row, n = nbase + ccp;
                if (crp < RPIP && row < WTMC && m < p.M && n < p.N) {
                    const f16x4 lo = *(const f16x4*)(wave_lds + row * LDWP + ccp);
                    const f16x4 hi = *(const f16x4*)(wave_lds + row * LDWP + ccp + 4);
                    *(f16x8*)(AUX + (int64_t)m * p.ldr + n) = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = nbase + j * 16 + lg * 4;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (p.bias && n < p.N) bv = *(const f32x4*)(p.bias + n);
#pragma unroll
            for (int ii = 0; ii < TMC; ++ii) {
                const f32x4 v = acc[c0 + ii][j] + bv;
                const f16 v0 = (f16)v[0], g0 = (f16)v[1], v1 = (f16)v[2], g1 = (f16)v[3];
                f16x2 o = {(f16)((float)v0 * gelu_erf_f((float)g0)), (f16)((float)v1 * gelu_erf_f((float)g1))};
                *(f16x2*)(wave_lds + (ii * 16 + l15) * LDW + j * 8 + lg * 2) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r0 = 0; r0 < WTMC; r0 += RPI) {
            const int row = r0 + cr;
            const int m = mbase + c0 * 16 + row, n = (nbase >> 1) + cc;
            if (cr < RPI && row < WTMC && m < p.M && n < No) {
                const f16x4 lo = *(const f16x4*)(wave_lds + row * LDW + cc);
                const f16x4 hi = *(const f16x4*)(wave_lds + row * LDW + cc + 4);
                *(f16x8*)((f16*)p.C + (int64_t)m * p.ldc + n) = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        }
        if (c0 + TMC < TM) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
}


#include <type_traits>
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int OFF>
__device__ __forceinline__ void ds_read16(f16x8& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void tie(f16x8& v) { asm volatile("" : "+v"(v)); }

// one k = 32 step of a wave tile: acc[i][j] += A_i . B_j^T over fragments at a_addr + i*2048 / b_addr + j*2048 (LDS byte addresses)
// GS: LDS bytes between consecutive 16-row fragment groups (2048 for 128-byte rows of the BK = 64 kernels, 1024 for BK = 32)
template <int TM, int TN, int PD, int GS = 2048>
__device__ __forceinline__ void mma_k32(f32x4 (&acc)[TM][TN], uint32_t a_addr, uint32_t b_addr) {
    constexpr bool BRES = TN <= TM;   // the operand with fewer fragments stays resident for the step
    constexpr int NR = BRES ? TN : TM, NS = BRES ? TM : TN, R = PD + 1;
    const uint32_t r_addr = BRES ? b_addr : a_addr, s_addr = BRES ? a_addr : b_addr;
    f16x8 res[NR], ring[R];
    static_for<0, NR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        ds_read16<i * GS>(res[i], r_addr);
    });
    static_for<0, (PD < NS ? PD : NS)>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        ds_read16<i * GS>(ring[i % R], s_addr);
    });
    static_for<0, NS>([&](auto ic) {
        constexpr int s = decltype(ic)::value;
        if constexpr (s + PD < NS) ds_read16<(s + PD) * GS>(ring[(s + PD) % R], s_addr);
        constexpr int after = (NS - 1 - s) < PD ? (NS - 1 - s) : PD;
        wait_lgkm<after>();
        if constexpr (s == 0) {
#pragma unroll
            for (int r = 0; r < NR; ++r) tie(res[r]);
        }
        tie(ring[s % R]);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if constexpr (BRES) acc[s][r] = FD_MFMA_16x16x32(res[r], ring[s % R], acc[s][r]);
            else acc[r][s] = FD_MFMA_16x16x32(ring[s % R], res[r], acc[r][s]);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

