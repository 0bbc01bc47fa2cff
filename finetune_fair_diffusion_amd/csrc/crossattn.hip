// The cross-attention sub-block of BasicTransformerBlock as ONE kernel (north_star's named fusion):
//
//     n2 = LayerNorm2(h1);  q = n2 . Wq^T;  o = softmax(q K^T / sqrt(d)) V  (K, V: the <= 80 prompt tokens of the sample's CFG half);
//     h2 = o . Wo^T + bo + h1;  n3 = LayerNorm3(h2)
//
// (diffusers attention.py BasicTransformerBlock.forward: norm2 -> attn2 -> residual -> norm3; attn2 is the attention processor selected at
// exp-1 main:811-817.)  Unfused, that is five launches -- fd_layernorm_fwd, fd_gemm, fd_attn_fwd, fd_gemm, fd_layernorm_fwd -- and n2, q, o and
// h2 each make a round trip through HBM.  Here a workgroup owns BM rows of the residual stream (64 rows and keeps them in ONE LDS tile [BM][C + 8] that is, in turn, n2, q, o and the staged h2:
//
//   P0  LayerNorm2 of the rows, one wave per row, same arithmetic as layernorm_kernel (identical statistics; outputs within one ulp), written to the tile;
//   P1  q = n2 . Wq^T: the four waves split the N = C columns (C / 4 each), so a B (weight) fragment is used by exactly one wave and goes
//       global -> registers directly (16 bytes per lane, prefetched two k-steps ahead); A fragments come from the tile; 16x16x32 MFMAs with the
//       operands swapped so that a lane holds q[m][n .. n + 3]; q is rounded once, multiplied by softmax_scale * log2(e), and overwrites the tile;
//   P2  attention: wave w owns heads 2w, 2w + 1.  S^T[key][query] = K_h . q_h^T (A = K rows from global / L2, B = q rows from the tile): a lane
//       holds ONE query and 4 keys per 16-key tile, so the softmax is in-lane plus two cross-lane steps, and the fp16 probabilities are directly
//       the B operand of O^T[dv][query] = V^T[dv][key] . P^T[key][query] (A = rows of the transposed V the caller prepared once per rollout; the
//       contraction's k-slots are key tiles (2j, 2j + 1) x 4 keys, the same order on both operands).  o overwrites the q columns of its head;
//   P3  h2 = o . Wo^T as P1; fp16(acc + bias) is staged in the tile, then one wave per row adds the residual h1 (fp16(staged + h1): the
//       rounding sequence of fd_gemm's LDS-staged epilogue, bit-identical h2 for identical o), stores h2, and LayerNorm3 of the row (as P0) -> n3.
//
// Algorithmic HBM traffic per row: read h1 twice (the second read is an L2 hit), write h2 and n3: 4 C * 2 B against 14 C * 2 B unfused.
// Template arguments: LORA -- the LoRA slabs of attn2.to_q / to_out ride in the kernel: t = tile . down^T (N = the padded rank, one 16-row MFMA tile per wave, rounded to
// the working dtype like fd_gemm's skinny kernel does) is parked in a second small LDS tile and enters the projection as one more k-step against up (the
// second K-slab of fd_gemm, same order); REC -- everything the backward consumes is written on the way: n2, the LayerNorm2 statistics, q as fd_attn_bwd_* take it, t_q,
// o, the log-sum-exp, t_o, the LayerNorm3 statistics (7 C * 2 B per row more; the recording forward of the finetuned model R1 = R3's forward, exp-1 main:1786-1795).
// The plain instantiation serves the frozen rollout R2 (exp-1 main:1844-1858).  The first transformer block of the U-Net keeps the separate launches (its query is
// computed once for the CFG pair), and so do the C = 1280 levels.
#include "common.h"

#define CA_LOG2E 1.4426950408889634f

template <int C> struct CrossCfg {
    static constexpr int BM = 64;                 // rows per workgroup (C = 640: 83 KB of LDS, one workgroup per CU)
    static constexpr int TM = BM / 16;            // 16-row MFMA tiles per wave (every wave covers all rows)
    static constexpr int WN = C / 4;              // output columns per wave
    static constexpr int TN = WN / 16;            // 16-column MFMA tiles per wave
    static constexpr int NK = C / 32;             // k-steps of the two projections
    static constexpr int PD = TN <= 5 ? 3 : 1;     // B-fragment prefetch distance (register sets: PD + 1)
    static constexpr int LDT = C + 8;             // LDS row stride (halfs): 16-byte aligned rows, conflict-free 16-byte fragment reads
    static constexpr int MAXV = (C / 8 + 63) / 64;                // 16-byte vectors per lane of a row
    static constexpr int D = C / 8;               // head dim (8 heads)
    static constexpr int NKS = (D + 31) / 32;     // k-steps of q . k^T
    static constexpr int NDT = (D + 15) / 16;     // 16-row tiles of O^T
    // workgroups per CU the register allocation is held to: two at C = 320 (253 registers, no scratch), so that one's row-wise phases and weight
    // latencies hide under the other's MFMAs; the wider tiles need more than 256 registers for their B-fragment sets (348 B of scratch when capped)
    static constexpr int OCC = C == 320 ? 2 : 1;
};

struct CrossArgs {
    const f16* x; const float* g2; const float* b2; float eps2;
    const f16* wq; const f16* k; const f16* vt; int Lp; int L;
    const f16* wo; const float* bo;
    const float* g3; const float* b3; float eps3;
    f16* y; f16* yn; float* yn_stats;
    int M, rows_per_sample, kv_div; float sl2;
    // LoRA slabs of attn2.to_q / to_out (LORA): down [rp, C] (row stride ld_*d), up [C, rp] (row stride ld_*u), rp <= 16
    const f16* qd; const f16* qu; const f16* od; const f16* ou; int ld_qd, ld_qu, ld_od, ld_ou, rp;
    // what the backward needs (REC): n2, LayerNorm2 statistics, q as the attention backward takes it (times q_store), t_q = n2 . down_q^T, o, the log-sum-exp, t_o
    f16* n2_out; float* ln2_stats; f16* q_out; f16* tq_out; f16* o_out; float* lse_out; f16* to_out; float q_store;
};

constexpr int CROSS_LDB = 40;        // row stride (halfs) of the LoRA t tile [64][32 + 8]: 16-byte aligned rows

// LayerNorm of R rows, each held as MAXV vectors per lane (vector v = lane + 64 i): per row the arithmetic of layernorm_kernel<false, .>, statement for
// statement (bit-identical outputs); the rows' shuffle trees are walked level by level so that R independent cross-lane exchanges are in flight instead of one
template <int R>
__device__ __forceinline__ void wave_sum_rows(float (&s)[R]) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float t[R];
#pragma unroll
        for (int r = 0; r < R; ++r) t[r] = __shfl_xor(s[r], o, 64);
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] += t[r];
    }
}
template <int C, int MAXV, int R>
__device__ __forceinline__ void ln_rows(const f16x8 (&xv)[R][MAXV], float eps, int lane, float (&mean)[R], float (&rstd)[R]) {
    constexpr int V = C / 8;
    float s0[R], s1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s0[r] = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int v = lane + i * 64;
            if (v < V) {
#pragma unroll
                for (int j = 0; j < 8; ++j) s0[r] += (float)xv[r][i][j];
            }
        }
    }
    wave_sum_rows<R>(s0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mean[r] = s0[r] / C;
        s1[r] = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int v = lane + i * 64;
            if (v < V) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = (float)xv[r][i][j] - mean[r];
                    s1[r] += d * d;
                }
            }
        }
    }
    wave_sum_rows<R>(s1);
#pragma unroll
    for (int r = 0; r < R; ++r) rstd[r] = rsqrtf(s1[r] / C + eps);
}
template <int MAXV>
__device__ __forceinline__ void ln_apply(const f16x8 (&xv)[MAXV], float mean, float rstd, const float (&gm)[MAXV][8], const float (&bt)[MAXV][8], f16x8 (&out)[MAXV]) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) out[i][j] = (f16)(((float)xv[i][j] - mean) * rstd * gm[i][j] + bt[i][j]);
}

// acc[i][j] (+)= tile rows [16 i, 16 i + 16) . W rows [n0 + 16 j, ...)^T over K = C; lane (l15, lg) ends up holding C[m = 16 i + l15][n = n0 + 16 j + 4 lg .. + 3]
// ``tb`` / ``up``: the LoRA slab t [64][CROSS_LDB] (LDS, zero beyond rp) against up [C, rp] -- one more k-step behind the main K, as fd_gemm's second K-slab
template <int C, bool LORA>
__device__ __forceinline__ void project(f32x4 (&acc)[CrossCfg<C>::TM][CrossCfg<C>::TN], const f16* __restrict__ tile, const f16* __restrict__ W, int n0, int l15,
                                        int lg, const f16* __restrict__ tb = nullptr, const f16* __restrict__ up = nullptr, int ld_up = 0, int rp = 0) {
    using Cf = CrossCfg<C>;
    constexpr int TM = Cf::TM, TN = Cf::TN, NK = Cf::NK, PD = Cf::PD, NS = PD + 1;
    f16x8 bf[NS][TN];
    const f16* wp = W + (int64_t)(n0 + l15) * C + lg * 8;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < PD; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[s][j] = *(const f16x8*)(wp + (int64_t)j * 16 * C + s * 32);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        if (ks + PD < NK) {
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[(ks + PD) % NS][j] = *(const f16x8*)(wp + (int64_t)j * 16 * C + (ks + PD) * 32);
        }
        f16x8 af[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *(const f16x8*)(tile + (i * 16 + l15) * Cf::LDT + ks * 32 + lg * 8);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = FD_MFMA_16x16x32(bf[ks % NS][j], af[i], acc[i][j]);
    }
    if constexpr (LORA) {
        f16x8 uf[TN], tf[TM];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            uf[j] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (lg * 8 < rp) uf[j] = *(const f16x8*)(up + (int64_t)(n0 + j * 16 + l15) * ld_up + lg * 8);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) tf[i] = *(const f16x8*)(tb + (i * 16 + l15) * CROSS_LDB + lg * 8);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = FD_MFMA_16x16x32(uf[j], tf[i], acc[i][j]);
    }
}

// t = tile . down^T (LoRA down-projection, N = rp <= 16): wave w owns rows [16 w, 16 w + 16); result rounded to the working dtype into tb (and t_out when recording)
template <int C>
__device__ __forceinline__ void lora_down(const f16* __restrict__ tile, const f16* __restrict__ down, int ld_down, int rp, f16* __restrict__ tb, f16* __restrict__ t_out,
                                          int64_t row0, int wave, int l15, int lg) {
    using Cf = CrossCfg<C>;
    static_assert(Cf::TM == 4, "one 16-row tile per wave");
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < Cf::NK; ++ks) {
        f16x8 df = {0, 0, 0, 0, 0, 0, 0, 0};
        if (l15 < rp) df = *(const f16x8*)(down + (int64_t)l15 * ld_down + ks * 32 + lg * 8);
        const f16x8 af = *(const f16x8*)(tile + (wave * 16 + l15) * Cf::LDT + ks * 32 + lg * 8);
        acc = FD_MFMA_16x16x32(df, af, acc);
    }
    // lane: t[row 16 wave + l15][columns 4 lg .. + 3]; columns >= rp are exact zeros (their down rows are)
    const f16x4 t4 = {(f16)acc[0], (f16)acc[1], (f16)acc[2], (f16)acc[3]};
    *(f16x4*)(tb + (wave * 16 + l15) * CROSS_LDB + lg * 4) = t4;
    if (t_out && lg * 4 < rp) *(f16x4*)(t_out + (row0 + wave * 16 + l15) * rp + lg * 4) = t4;
}

template <int C, bool LORA, bool REC>
__global__ __launch_bounds__(256, CrossCfg<C>::OCC) void cross_block_kernel(CrossArgs a) {
    FD_WG_TRACE(10);
    using Cf = CrossCfg<C>;
    constexpr int BM = Cf::BM, TM = Cf::TM, TN = Cf::TN, LDT = Cf::LDT, MAXV = Cf::MAXV, D = Cf::D, NKS = Cf::NKS, NDT = Cf::NDT, V = C / 8;
    constexpr int RPW = BM / 4;                   // rows per wave in the row-wise phases
    constexpr int RG = RPW * MAXV <= 16 ? RPW : 16 / MAXV;      // ... taken in groups of RG rows (<= 16 vectors = 64 registers of row data per lane at a time)
    extern __shared__ __attribute__((aligned(16))) f16 tile[];       // [BM][LDT], then (LORA) the t tile [BM][CROSS_LDB]
    f16* const tb = tile + BM * LDT;
    if (LORA) {          // columns 16 .. 39 are never written again: zero once (columns < 16 are written by lora_down before their first read)
        for (int c = threadIdx.x; c < BM * CROSS_LDB; c += 256) tb[c] = (f16)0.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int row0 = blockIdx.x * BM;
    const int bk = (row0 / a.rows_per_sample) / a.kv_div;

    // ---------------------------------------------------------------- P0: n2 = LayerNorm2(h1) -> tile
    {
        float gm[MAXV][8], bt[MAXV][8];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int v = lane + i * 64;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                gm[i][j] = v < V ? a.g2[v * 8 + j] : 0.f;
                bt[i][j] = v < V ? a.b2[v * 8 + j] : 0.f;
            }
        }
#pragma unroll 1
        for (int rg = 0; rg < RPW; rg += RG) {
            f16x8 xv[RG][MAXV];
    #pragma unroll
            for (int r = 0; r < RG; ++r)
    #pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int v = lane + i * 64;
                    xv[r][i] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    if (v < V) xv[r][i] = *(const f16x8*)(a.x + (int64_t)(row0 + wave * RPW + rg + r) * C + v * 8);
                }
            float mean[RG], rstd[RG];
            ln_rows<C, MAXV, RG>(xv, a.eps2, lane, mean, rstd);
    #pragma unroll
            for (int r = 0; r < RG; ++r) {
                f16x8 o[MAXV];
                ln_apply<MAXV>(xv[r], mean[r], rstd[r], gm, bt, o);
    #pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int v = lane + i * 64;
                    if (v < V) {
                        *(f16x8*)(tile + (wave * RPW + rg + r) * LDT + v * 8) = o[i];
                        if (REC) *(f16x8*)(a.n2_out + (int64_t)(row0 + wave * RPW + rg + r) * C + v * 8) = o[i];
                    }
                }
                if (REC && lane == 0) {
                    a.ln2_stats[(int64_t)(row0 + wave * RPW + rg + r) * 2] = mean[r];
                    a.ln2_stats[(int64_t)(row0 + wave * RPW + rg + r) * 2 + 1] = rstd[r];
                }
            }
        }
    }
    __syncthreads();
    if (LORA) {
        lora_down<C>(tile, a.qd, a.ld_qd, a.rp, tb, REC ? a.tq_out : nullptr, row0, wave, l15, lg);
        __syncthreads();
    }

    // ---------------------------------------------------------------- P1: q = n2 . Wq^T, scaled into the exponent's domain -> tile
    f32x4 acc[TM][TN];
    const int n0 = wave * Cf::WN;
#ifdef FD_CROSS_SKIP        // measurement builds only (scratch/r05_passes.sh o): phases left out, results meaningless
    constexpr int SKIP = FD_CROSS_SKIP;
#else
    constexpr int SKIP = 0;
#endif
    if (!(SKIP & 2)) project<C, LORA>(acc, tile, a.wq, n0, l15, lg, tb, a.qu, a.ld_qu, a.rp);
    else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){1.f, 1.f, 1.f, 1.f};
    }
    __syncthreads();                               // every wave has read all of n2
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const f32x4 v = acc[i][j] * a.sl2;
            const f16x4 q4 = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
            *(f16x4*)(tile + (i * 16 + l15) * LDT + n0 + j * 16 + lg * 4) = q4;
            if (REC && a.q_store != a.sl2) {     // the backward takes q unscaled (head dims without spare contraction slots): its own rounding, stored from the registers
                const f32x4 w = acc[i][j] * a.q_store;
                *(f16x4*)(a.q_out + (int64_t)(row0 + i * 16 + l15) * C + n0 + j * 16 + lg * 4) = (f16x4){(f16)w[0], (f16)w[1], (f16)w[2], (f16)w[3]};
            }
        }
    if (REC && a.q_store == a.sl2) {             // the tile IS the backward's q: whole rows, 16 bytes per lane (two workgroup barriers instead of 8-byte scattered stores)
        __syncthreads();
#pragma unroll 4
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int v = lane + i * 64;
                if (v < V) *(f16x8*)(a.q_out + (int64_t)(row0 + wave * RPW + r) * C + v * 8) = *(const f16x8*)(tile + (wave * RPW + r) * LDT + v * 8);
            }
        __syncthreads();
    }
    // P2 reads only the columns this wave wrote (its two heads: 2 D = C / 4 columns): wave-local ordering is enough
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    // ---------------------------------------------------------------- P2: attention over the <= 80 keys, heads 2 wave, 2 wave + 1; o overwrites q
#pragma unroll 1
    for (int hh = 0; hh < ((SKIP & 4) ? 0 : 2); ++hh) {
        const int c0 = (wave * 2 + hh) * D;       // first column of the head
        // K fragments [key tile][k-step]: lane (key = 16 kt + l15, k = 32 ks + 8 lg ..); k >= D and keys >= L are zero
        f16x8 kf[5][NKS];
#pragma unroll
        for (int kt = 0; kt < 5; ++kt)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int key = kt * 16 + l15, kk = ks * 32 + lg * 8;
                kf[kt][ks] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (key < a.L && kk < D) kf[kt][ks] = *(const f16x8*)(a.k + ((int64_t)bk * a.L + key) * C + c0 + kk);
            }
        // V^T fragments [dv tile][key-tile pair]: lane (dv = 16 dt + l15; keys 16 (2 jp) + 4 lg .. + 3 and 16 (2 jp + 1) + 4 lg .. + 3); rows dv >= D are
        // another head's (or clamped): they only feed output rows that are never stored
        f16x8 vf[NDT][3];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const int dv = min(c0 + dt * 16 + l15, C - 1);
            const f16* vp = a.vt + ((int64_t)bk * C + dv) * a.Lp + lg * 4;
#pragma unroll
            for (int jp = 0; jp < 3; ++jp) {
                const f16x4 lo = *(const f16x4*)(vp + jp * 32);
                f16x4 hi = {0, 0, 0, 0};
                if (jp < 2) hi = *(const f16x4*)(vp + jp * 32 + 16);
                vf[dt][jp] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        }
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            f16x8 qf[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int kk = ks * 32 + lg * 8;
                qf[ks] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (kk < D) qf[ks] = *(const f16x8*)(tile + (mt * 16 + l15) * LDT + c0 + kk);
            }
            f32x4 s[5];
#pragma unroll
            for (int kt = 0; kt < 5; ++kt) {
                s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) s[kt] = FD_MFMA_16x16x32(kf[kt][ks], qf[ks], s[kt]);
            }
            // lane: query mt * 16 + l15, keys 16 kt + 4 lg + r
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (kt * 16 + lg * 4 + r >= a.L) s[kt][r] = -INFINITY;
                    mx = fmaxf(mx, s[kt][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            f16x4 p16[6];
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p16[kt][r] = (f16)__builtin_amdgcn_exp2f(s[kt][r] - mx);
                    l += (float)p16[kt][r];
                }
            p16[5] = (f16x4){0, 0, 0, 0};
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            const float inv = 1.f / l;
            if (REC && lg == 0) {      // natural-log sum-exp of the scaled scores, as fd_attn_fwd writes it: [B, H, rows_per_sample]
                const int64_t row = row0 + mt * 16 + l15;
                const int64_t smp = row / a.rows_per_sample;
                a.lse_out[(smp * 8 + (wave * 2 + hh)) * a.rows_per_sample + (row - smp * a.rows_per_sample)] = (mx + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
            }
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jp = 0; jp < 3; ++jp) {
                    const f16x8 pb = {p16[2 * jp][0], p16[2 * jp][1], p16[2 * jp][2], p16[2 * jp][3],
                                      p16[2 * jp + 1][0], p16[2 * jp + 1][1], p16[2 * jp + 1][2], p16[2 * jp + 1][3]};
                    o = FD_MFMA_16x16x32(vf[dt][jp], pb, o);
                }
                // lane: o[query l15][dv = 16 dt + 4 lg + r]
                if (dt * 16 + lg * 4 < D) {
                    const f16x4 o4 = {(f16)(o[0] * inv), (f16)(o[1] * inv), (f16)(o[2] * inv), (f16)(o[3] * inv)};
                    *(f16x4*)(tile + (mt * 16 + l15) * LDT + c0 + dt * 16 + lg * 4) = o4;
                }
            }
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------- P3: h2 = o . Wo^T + bo + h1 -> y;  n3 = LayerNorm3(h2) -> yn
    if (REC) {                                   // o for the backward: whole rows of the tile (nothing writes it before the barrier behind the projection)
#pragma unroll 4
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int v = lane + i * 64;
                if (v < V) *(f16x8*)(a.o_out + (int64_t)(row0 + wave * RPW + r) * C + v * 8) = *(const f16x8*)(tile + (wave * RPW + r) * LDT + v * 8);
            }
    }
    if (LORA) {
        lora_down<C>(tile, a.od, a.ld_od, a.rp, tb, REC ? a.to_out : nullptr, row0, wave, l15, lg);
        __syncthreads();
    }
    if (!(SKIP & 8)) project<C, LORA>(acc, tile, a.wo, n0, l15, lg, tb, a.ou, a.ld_ou, a.rp);
    __syncthreads();                               // every wave has read all of o
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const f32x4 bv = *(const f32x4*)(a.bo + n0 + j * 16 + lg * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const f32x4 v = acc[i][j] + bv;
            *(f16x4*)(tile + (i * 16 + l15) * LDT + n0 + j * 16 + lg * 4) = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
        }
    }
    __syncthreads();
    {
        float gm[MAXV][8], bt[MAXV][8];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int v = lane + i * 64;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                gm[i][j] = (a.yn && v < V) ? a.g3[v * 8 + j] : 0.f;
                bt[i][j] = (a.yn && v < V) ? a.b3[v * 8 + j] : 0.f;
            }
        }
#pragma unroll 1
        for (int rg = 0; rg < RPW; rg += RG) {
            f16x8 rv[RG][MAXV];
    #pragma unroll
            for (int r = 0; r < RG; ++r)
    #pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int v = lane + i * 64;
                    rv[r][i] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    if (v < V) rv[r][i] = *(const f16x8*)(a.x + (int64_t)(row0 + wave * RPW + rg + r) * C + v * 8);
                }
            f16x8 hv[RG][MAXV];
    #pragma unroll
            for (int r = 0; r < RG; ++r) {
                const int64_t row = row0 + wave * RPW + rg + r;
    #pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int v = lane + i * 64;
                    hv[r][i] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    if (v < V) {
                        const f16x8 st = *(const f16x8*)(tile + (wave * RPW + rg + r) * LDT + v * 8);
    #pragma unroll
                        for (int j = 0; j < 8; ++j) hv[r][i][j] = (f16)((float)st[j] + (float)rv[r][i][j]);
                        *(f16x8*)(a.y + row * C + v * 8) = hv[r][i];
                    }
                }
            }
            if (a.yn) {
                float mean[RG], rstd[RG];
                ln_rows<C, MAXV, RG>(hv, a.eps3, lane, mean, rstd);
    #pragma unroll
                for (int r = 0; r < RG; ++r) {
                    const int64_t row = row0 + wave * RPW + rg + r;
                    f16x8 o[MAXV];
                    ln_apply<MAXV>(hv[r], mean[r], rstd[r], gm, bt, o);
    #pragma unroll
                    for (int i = 0; i < MAXV; ++i) {
                        const int v = lane + i * 64;
                        if (v < V) *(f16x8*)(a.yn + row * C + v * 8) = o[i];
                    }
                    if (a.yn_stats && lane == 0) {
                        a.yn_stats[row * 2] = mean[r];
                        a.yn_stats[row * 2 + 1] = rstd[r];
                    }
                }
            }
        }
    }
}

extern "C" int fd_cross_attn_block(const fd_cross_block_desc* dp, void* stream) {
    FD_REQUIRE_DESC(dp, fd_cross_block_desc, "fd_cross_attn_block");
    const fd_cross_block_desc& d = *dp;
    FD_REQUIRE(d.x && d.ln2_gamma && d.ln2_beta && d.wq && d.k && d.vt && d.wo && d.bo && d.y, "fd_cross_attn_block: null operand");
    FD_REQUIRE(!d.yn || (d.ln3_gamma && d.ln3_beta), "fd_cross_attn_block: yn needs ln3_gamma / ln3_beta");
    // C = 1280 (the 16^2 / 8^2 levels, M <= 4096 rows) is refused: a 64-row tile of 1280 columns does not fit the LDS, and with 16-row tiles every workgroup
    // streams both 3.3 MB weight matrices for 16 rows -- measured 205 us against 103 us for the five launches (profiles/r05_cross_block_fused.txt)
    FD_REQUIRE(d.C == 320 || d.C == 640, "fd_cross_attn_block: C=%d (320 / 640: eight heads of 40 / 80)", d.C);
    FD_REQUIRE(d.heads == 8, "fd_cross_attn_block: heads=%d (8)", d.heads);
    const int bm = 64;
    FD_REQUIRE(d.M > 0 && d.M % bm == 0 && d.rows_per_sample > 0 && d.rows_per_sample % bm == 0 && d.M % d.rows_per_sample == 0,
               "fd_cross_attn_block: M=%d and rows_per_sample=%d must be multiples of the %d-row tile", d.M, d.rows_per_sample, bm);
    FD_REQUIRE(d.L > 0 && d.L <= 80 && d.Lp >= 80 && (d.Lp & 3) == 0 && d.kv_div >= 1 && (d.M / d.rows_per_sample) % d.kv_div == 0,
               "fd_cross_attn_block: L=%d (<= 80 keys), Lp=%d (>= 80, key-padded rows of vt), kv_div=%d", d.L, d.Lp, d.kv_div);
    CrossArgs a;
    a.x = (const f16*)d.x; a.g2 = d.ln2_gamma; a.b2 = d.ln2_beta; a.eps2 = d.ln2_eps;
    a.wq = (const f16*)d.wq; a.k = (const f16*)d.k; a.vt = (const f16*)d.vt; a.Lp = d.Lp; a.L = d.L;
    a.wo = (const f16*)d.wo; a.bo = d.bo; a.g3 = d.ln3_gamma; a.b3 = d.ln3_beta; a.eps3 = d.ln3_eps;
    a.y = (f16*)d.y; a.yn = (f16*)d.yn; a.yn_stats = d.yn_stats;
    a.M = d.M; a.rows_per_sample = d.rows_per_sample; a.kv_div = d.kv_div;
    a.sl2 = d.scale * CA_LOG2E;
    const bool lora = d.lora_q_down != nullptr, rec = d.n2_out != nullptr;
    FD_REQUIRE(!lora || (d.lora_q_up && d.lora_o_down && d.lora_o_up && d.lora_rp > 0 && d.lora_rp <= 16 && (d.lora_rp & 7) == 0 &&
                         (d.ld_q_down & 7) == 0 && (d.ld_q_up & 7) == 0 && (d.ld_o_down & 7) == 0 && (d.ld_o_up & 7) == 0),
               "fd_cross_attn_block: LoRA slabs need all four matrices, a padded rank of 8 or 16 (got %d) and row strides that are multiples of 8", d.lora_rp);
    FD_REQUIRE(!rec || (d.ln2_stats && d.q_out && d.o_out && d.lse_out && (!lora || (d.tq_out && d.to_out))),
               "fd_cross_attn_block: recording needs n2_out, ln2_stats, q_out, o_out, lse_out (and tq_out, to_out with LoRA slabs)");
    a.qd = (const f16*)d.lora_q_down; a.qu = (const f16*)d.lora_q_up; a.od = (const f16*)d.lora_o_down; a.ou = (const f16*)d.lora_o_up;
    a.ld_qd = (int)d.ld_q_down; a.ld_qu = (int)d.ld_q_up; a.ld_od = (int)d.ld_o_down; a.ld_ou = (int)d.ld_o_up; a.rp = d.lora_rp;
    a.n2_out = (f16*)d.n2_out; a.ln2_stats = d.ln2_stats; a.q_out = (f16*)d.q_out; a.tq_out = (f16*)d.tq_out; a.o_out = (f16*)d.o_out; a.lse_out = d.lse_out;
    a.to_out = (f16*)d.to_out;
    a.q_store = d.q_prescaled ? a.sl2 : 1.f;
    const dim3 grid(d.M / bm), block(256);
#define CROSS_LAUNCH(CC, LL, RR)                                                                                                        \
    {                                                                                                                                   \
        constexpr size_t lds = ((size_t)CrossCfg<CC>::BM * CrossCfg<CC>::LDT + (LL ? CrossCfg<CC>::BM * CROSS_LDB : 0)) * 2;            \
        static bool once = false;                                                                                                       \
        if (!once) {                                                                                                                    \
            (void)hipFuncSetAttribute((const void*)cross_block_kernel<CC, LL, RR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            once = true;                                                                                                                \
        }                                                                                                                               \
        hipLaunchKernelGGL((cross_block_kernel<CC, LL, RR>), grid, block, lds, (hipStream_t)stream, a);                                 \
    }
#define CROSS_LAUNCH_C(CC)                                       \
    {                                                            \
        if (lora && rec) CROSS_LAUNCH(CC, true, true)            \
        else if (lora) CROSS_LAUNCH(CC, true, false)             \
        else if (rec) CROSS_LAUNCH(CC, false, true)              \
        else CROSS_LAUNCH(CC, false, false)                      \
    }
    if (d.C == 320) CROSS_LAUNCH_C(320) else CROSS_LAUNCH_C(640)
#undef CROSS_LAUNCH_C
#undef CROSS_LAUNCH
    return fd_check_launch("fd_cross_attn_block");
}

FD_WGT_SETTER(crossattn)
