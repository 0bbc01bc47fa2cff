// Fused (flash-style) attention for gfx950: forward, dQ and dK/dV kernels.
//
// All three kernels use v_mfma_f32_32x32x16_f16 in the "transposed score" arrangement:
//   S^T[key][q] = K . Q^T   (A = K rows from LDS, B = Q rows held in VGPRs)
// so a lane owns ONE query column (q = lane & 31) and 16 keys per 32-key sub-tile
// (key = (r&3) + 8*(r>>2) + 4*(lane>>5)); the softmax row reduction is in-lane plus one
// cross-half shuffle, and P (fp16) is directly the B operand of the next MFMA
//   O^T[dv][q] += V^T[dv][key] . P^T[key][q]
// with the k index of that MFMA permuted consistently on both operands (element j of a lane's
// 8-wide operand <-> key 4*(lane>>5) + (j&3) + 8*(j>>2) of the 16-key step), which the A side
// realises as transpose reads (ds_read_b64_tr_b16) of the row-major tile the projections wrote.  Head dims that are
// not MFMA multiples (40, 80) are zero-padded in LDS: D -> DK (x16) for contractions over d and
// D -> DV (x32) where d is an output dimension.
#include "common.h"

#define LOG2E 1.4426950408889634f
#ifndef FD_ATTN_LAZY
#define FD_ATTN_LAZY 8.0f      // 0 = move the softmax reference point with every new maximum (rounds 1-3)
#endif

__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return FD_MFMA_32x32x16(a, b, c); }
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
// Workgroup -> (batch, head, 128-row block).  Launches are 1-D; consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with its
// own 4 MB L2.  All row blocks of one (batch, head) re-read that head's K / V (0.66 MB at T = 4096, d = 40), so they are given ids that are
// congruent mod 8: an XCD then holds the K / V of the few heads it is working on instead of every XCD holding every head's.
__device__ __forceinline__ void attn_block_coords(int nblk, int H, int B, int& b, int& h, int& blk) {
    const int L = blockIdx.x, NH = H * B;
#ifndef FD_ATTN_NO_XCD_MAP
    if ((NH & 7) == 0) {
        const int xcd = L & 7, m = L >> 3;
        blk = m % nblk;
        const int hh = (m / nblk) * 8 + xcd;
        b = hh / H;
        h = hh - b * H;
        return;
    }
#endif
    blk = L % nblk;
    const int hh = L / nblk;
    b = hh / H;
    h = hh - b * H;
}

// row index inside a 32x32 C/D tile for accumulator register r of lane-half g
__device__ __forceinline__ int crow(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

// Row padding (halfs) of the backward kernels' row-major Q / dO / K / V tiles, read both as 16-byte fragments and through read_tr.
// Measurement knob (make BENCH_HOOKS=1 EXTRA_DEFS=-DFD_ATTN_BWD_PAD=n): 8 is the shipped value.
#ifndef FD_ATTN_BWD_PAD
#define FD_ATTN_BWD_PAD 8
#endif
// The permuted-k A operand of a product whose A matrix is the TRANSPOSE of a row-major LDS tile [k][m] (row stride ``stride``
// halfs, 8-byte aligned rows): gfx950's ds_read_b64_tr_b16.  A 16-lane group fetches one [4 k][16 m] block -- lane L of the group supplies
// the address of row L / 4, columns 4 (L % 4) .. + 3 -- and lane c receives column c, i.e. A[m0 + c][k .. k + 3].  With it V (forward),
// K (dQ) and Q / dO (dK, dV) are consumed in the layout the projections write them in: no transposed copies in HBM, no transposed tiles.
// Columns past the tile's row only feed output rows >= D, which are never stored.
typedef short fd_s16x4 __attribute__((__vector_size__(8)));
__device__ __forceinline__ f16x8 read_tr(const f16* tile, int stride, int k0, int m0, int ml, int g) {
    const f16* p = tile + (k0 + 4 * g + ((ml & 15) >> 2)) * stride + m0 + (ml & 16) + 4 * (ml & 3);
    const fd_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fd_s16x4*)p);
    const fd_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fd_s16x4*)(p + 8 * stride));
    const f16x4 a = __builtin_bit_cast(f16x4, lo), b = __builtin_bit_cast(f16x4, hi);
    return (f16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
// row stride (halfs) of a tile that is only read with read_tr: the 32 lanes serviced together fetch 4 rows x 64 bytes, which sit on distinct
// banks when the stride is 64 or 192 bytes modulo 256
constexpr int tr_stride(int DV) { return DV % 128 == 32 || DV % 128 == 96 ? DV : (DV + 32) % 128 == 32 || (DV + 32) % 128 == 96 ? DV + 32 : DV + 64; }


// ---- register-staged tiles with the load split from the LDS write (issue the next tile's global loads before
// computing the current one, write them to LDS after the barrier): hides the HBM/L2 latency under the MFMAs.
template <int D> struct TileRegs { f16x8 r[(64 * (D / 8) + 255) / 256]; };

template <int D>
__device__ __forceinline__ void load_rows(TileRegs<D>& t, const f16* src, int64_t ld, int row0, int nrows_valid) {
    constexpr int CH = D / 8, N = 64 * CH, NCH = (N + 255) / 256;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = threadIdx.x + i * 256;
        const int r = c / CH, cc = (c - r * CH) * 8;
        t.r[i] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (c < N && row0 + r < nrows_valid) t.r[i] = *(const f16x8*)(src + (int64_t)(row0 + r) * ld + cc);
    }
}
// Interior tiles (all 64 rows valid) need no bounds logic: per-thread byte offsets are computed once per kernel and a load is
// uniform base + 32-bit offset.  The generic versions above/below remain for the last, partial tile.  Used by the forward kernel only:
// in the backward kernels the extra offset registers cost an occupancy step (dK/dV at d=40: 250 -> 256 VGPRs, 27 % slower).
template <int D> struct TilePlan { unsigned off[(64 * (D / 8) + 255) / 256]; };
template <int D>
__device__ __forceinline__ void plan_rows(TilePlan<D>& pl, int64_t ld) {
    constexpr int CH = D / 8, N = 64 * CH, NCH = (N + 255) / 256;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = min((int)threadIdx.x + i * 256, N - 1);   // threads beyond the tile re-read its last chunk (never stored)
        const int r = c / CH, cc = (c - r * CH) * 8;
        pl.off[i] = (unsigned)((r * ld + cc) * 2);
    }
}
template <int D>
__device__ __forceinline__ void load_planned(TileRegs<D>& t, const f16* tile_base /* wave-uniform */, const TilePlan<D>& pl) {
    constexpr int NCH = (64 * (D / 8) + 255) / 256;
#pragma unroll
    for (int i = 0; i < NCH; ++i) t.r[i] = *(const f16x8*)((const char*)tile_base + pl.off[i]);
}
template <int D, int DKP>
__device__ __forceinline__ void store_rows(const TileRegs<D>& t, f16* dst) {
    constexpr int CH = D / 8, N = 64 * CH, NCH = (N + 255) / 256;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = threadIdx.x + i * 256;
        const int r = c / CH, cc = (c - r * CH) * 8;
        if (c < N) *(f16x8*)(dst + r * DKP + cc) = t.r[i];
    }
}
// zero the padding that staging never touches: columns [D, DKP) of a row tile
template <int D, int DKP>
__device__ __forceinline__ void zero_row_pad(f16* dst) {
    for (int c = threadIdx.x; c < 64 * (DKP - D); c += 256) dst[(c / (DKP - D)) * DKP + D + c % (DKP - D)] = (f16)0;
}
// Occupancy targets (waves per SIMD, the second __launch_bounds__ argument).  Without one the register allocator spreads out to whatever
// the 4-wave workgroup allows (512 per lane): round 1 shipped 184 (forward), 202 (dQ) and 310 (dK/dV) VGPR+AGPR at d = 40, i.e. 2 / 2 / 1
// waves per SIMD, while reading only the arch-VGPR half of that number as "fits three".  The values below are the highest occupancy
// each head-dim class reaches WITHOUT scratch (checked on the gfx950 ISA: .amdhsa_next_free_vgpr / private_segment_fixed_size).
constexpr int fwd_waves(int D) { return D <= 64 ? 3 : D <= 128 ? 2 : 1; }
constexpr int dq_waves(int D) { return D <= 40 ? 3 : D <= 128 ? 2 : 1; }
#ifdef FD_DKDV_TR_W3      // measurement: dK/dV at d = 40 squeezed to three waves per SIMD (168 registers + 124 B of scratch)
constexpr int dkdv_waves(int D) { return D == 40 ? 3 : D <= 64 ? 2 : 1; }
#else
constexpr int dkdv_waves(int D) { return D <= 64 ? 2 : 1; }
#endif

// ---- "-D through the matrix pipe" (round 4).  dS = P o (dP - D), D = rowsum(dO o O): where the contraction over d is padded (d = 40 -> 48: eight spare
// k-slots behind column 39) the subtraction rides in the dP = dO . V^T MFMAs: three slots of the dO operand carry -D split into three 16-bit
// pieces at the scales 256, 1, 1 / 256 (split3_scaled: |D| up to 1.6e7 stays inside the fp16 range, no piece is ever subnormal), the
// same three slots of the V operand carry those scales -- products and accumulation are exact in the fp32 accumulator, so dP' = dP - D comes out of the
// accumulation itself and the VALU-bound score loop loses one of its ~7.5 issue slots per element (and the dK/dV kernel its per-element LDS read of D).
// ---- "pre-scaled q" (round 4; negative ``scale`` argument of the three entry points).  The projection that writes q multiplies it by softmax_scale *
// log2(e) in its fp32 epilogue (fd_gemm_desc.colscale: one rounding, as before), so the QK^T accumulator IS the exponent's argument up to the
// reference point -- and the reference point (the running maximum in the forward, the saved log-sum-exp in the backward) rides in the same three
// spare contraction slots as -D does: q's slots carry -m or -lse split into three 16-bit pieces, K's slots carry 1.0.  p = exp2(accumulator): the
// fused multiply-add per score element is gone from all three kernels.  Head dims with the slots only (d = 40).
template <int D> constexpr bool pre_ok() { return D % 16 == 8; }
#define FD_PRE_MASKED 30000.f        // "lse" of an invalid query row in the pre-scaled backward: exp2(s - 30000) == 0, and it splits into finite pieces

template <int D> constexpr bool dfold() {
#ifdef FD_ATTN_NO_DFOLD
    return false;
#else
    return D % 16 == 8;      // lane-half 1 of the last k-step holds columns D .. D + 7: all padding
#endif
}
__device__ __forceinline__ void split3(float x, f16& h0, f16& h1, f16& h2) {
    h0 = (f16)x;
    const float r1 = x - (float)h0;
    h1 = (f16)r1;
    h2 = (f16)(r1 - (float)h1);
}
// -D for the dO . V^T product: three pieces at three SCALES -- x = 256 h0 + h1 + h2 / 256 against V slots (256, 1, 1/256) -- so that every piece
// is either a NORMAL 16-bit number or exactly zero, whatever |D| is.  The round-4 form carried x / 256 in all three slots: under loss scaling
// (dO ~ 1e-3 .. 1e-4 at the 64^2 level) |D| drops below 256 * 6.1e-5 = 0.016, the leading piece became an fp16 subnormal (8e-6 absolute error floor on D, and
// all of D lost if the matrix pipe flushes subnormal inputs).  Here a piece that would be subnormal is replaced by zero and its value moves to the next,
// finer slot: absolute error floor 6.1e-5 / 256 * 2^-11 ~ 1e-10 down to |D| ~ 2.4e-7, below which D is dropped (nothing in a scaled backward is that small
// and still matters).  bf16 has the fp32 exponent range: the flush branches never fire and the pieces carry 3 x 8 bits as before.
#ifdef FD_BF16
#define FD_WD_MIN_NORMAL 1.1754944e-38f
#else
#define FD_WD_MIN_NORMAL 6.103515625e-5f
#endif
__device__ __forceinline__ void split3_scaled(float x, f16& h0, f16& h1, f16& h2) {
    const float x0 = x * (1.f / 256.f);
    h0 = fabsf(x0) >= FD_WD_MIN_NORMAL ? (f16)x0 : (f16)0.f;
    const float r1 = x - 256.f * (float)h0;                          // exact: the product has 11 significant bits
    h1 = fabsf(r1) >= FD_WD_MIN_NORMAL ? (f16)r1 : (f16)0.f;
    const float r2 = (r1 - (float)h1) * 256.f;
    h2 = fabsf(r2) >= FD_WD_MIN_NORMAL ? (f16)r2 : (f16)0.f;
}

// ================================================================================== forward
// V is read as the projection wrote it ([Bk, Tkr, .] rows of stride ldk, like K): the PV operand comes from its row-major tile through read_tr.
// (Rounds 1-2 read a transposed copy V^T made by fd_transpose_btc; that form left the product in round 5 -- git history, profiles/r03_attention_transpose_read_ab.txt.)
// QB: 32-query column blocks per wave.  QB = 1: 4 waves x 32 queries per workgroup (rounds 1-3).  QB = 2 (round 4): a wave owns 64 consecutive
// queries, so every K / V fragment it reads from LDS feeds two MFMAs and the staged K / V tile serves 256 queries instead of 128 -- both the
// fragment traffic and the staging per query halve (the forward was co-limited by exactly those: MFMAs + fragment reads alone 410 of 646 us,
// staging + barriers ~190, profiles/r02_attn_fwd_d40_ablation.txt), at two waves per SIMD instead of three.
template <int D, int QB, bool PRE = false>
__global__ __launch_bounds__(256, QB == 1 ? fwd_waves(D) : (D <= 64 ? 2 : 1)) void attn_fwd_kernel(const f16* __restrict__ Q, const f16* __restrict__ K, const f16* __restrict__ V,
                                                       f16* __restrict__ O, float* __restrict__ LSE, int H, int Tq, int Tk,
                                                       int Tkr, int kv_div, float scale, int ldq, int ldk) {
    FD_WG_TRACE(7);
    constexpr int DK = (D + 15) / 16 * 16, DV = (D + 31) / 32 * 32, DKP = DK + 8;
    constexpr int NKS = DK / 16, NDV = DV / 32;
    constexpr int RB = 128 * QB;                // query rows per workgroup
    extern __shared__ __attribute__((aligned(16))) f16 smem[];
    constexpr int VP = tr_stride(DV);
    f16* Ks = smem;               // [64][DKP]
    f16* Vts = smem + 64 * DKP;   // [64][VP], keys x d

    int b, h, qblk;
    attn_block_coords((Tq + RB - 1) / RB, H, gridDim.x / (((Tq + RB - 1) / RB) * H), b, h, qblk);
    const int q0 = qblk * RB;
    const int bk = b / kv_div;
    const int C = H * D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 31, g = lane >> 5;
    int t[QB];
    bool tvalid[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        t[qb] = q0 + (wave * QB + qb) * 32 + ql;
        tvalid[qb] = t[qb] < Tq;
    }

    f16x8 qf[QB][NKS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int col = ks * 16 + g * 8;
            qf[qb][ks] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (tvalid[qb] && col < D) qf[qb][ks] = *(const f16x8*)(Q + ((int64_t)b * Tq + t[qb]) * ldq + h * D + col);
        }
    f32x16 oacc[QB][NDV];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int i = 0; i < NDV; ++i) oacc[qb][i] = zero16();
        m_run[qb] = PRE ? 0.f : -INFINITY;      // PRE: a finite reference point in the log2 domain (q's slots hold -m_run: zero to start with)
        l_run[qb] = 0.f;
    }
    static_assert(!PRE || pre_ok<D>(), "pre-scaled q: head dims with three spare contraction slots");
    const float sl2 = scale * LOG2E;

    const f16* Kb = K + (int64_t)bk * Tkr * ldk + h * D;
    const f16* Vtb = V + (int64_t)bk * Tkr * ldk + h * D;

    TileRegs<D> kreg, vreg;
    TilePlan<D> kplan, vplan;
    plan_rows<D>(kplan, ldk);
    plan_rows<D>(vplan, ldk);
    zero_row_pad<D, DKP>(Ks);
    // Head dims with a spare padded output row (40, 80, 16): "V column D" is a column of ones, so row D of O^T accumulates sum_k p -- the softmax
    // denominator comes out of the P.V MFMAs (with the same rescaling as O) instead of 32 VALU adds and a shuffle per tile; it is the sum of
    // the fp16-rounded probabilities the numerator is built from.
#ifdef FD_ATTN_NO_ONES
    constexpr bool ONES = false;
#else
    constexpr bool ONES = D < DV;
#endif
    if (PRE) {
        __syncthreads();                               // behind zero_row_pad's writes of the same columns
        for (int c = threadIdx.x; c < 64 * 3; c += 256) Ks[(c / 3) * DKP + D + c % 3] = (f16)1.f;   // never overwritten: store_rows writes columns < D
    }
    if (ONES) {
        for (int r = threadIdx.x; r < 64; r += 256) Vts[r * VP + D] = (f16)1.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // Q fragments landed: no VM event may pend on them inside the loop
    load_rows<D>(kreg, Kb, ldk, 0, Tk);
    load_rows<D>(vreg, Vtb, ldk, 0, Tk);
    for (int k0 = 0; k0 < Tk; k0 += 64) {
        __syncthreads();
        store_rows<D, DKP>(kreg, Ks);
        store_rows<D, VP>(vreg, Vts);
        __syncthreads();
        f32x16 s[QB][2];
        auto scores = [&]() {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) s[qb][kt] = zero16();
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const f16x8 kf = *(const f16x8*)(Ks + (kt * 32 + ql) * DKP + ks * 16 + g * 8);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) s[qb][kt] = mfma32(kf, qf[qb][ks], s[qb][kt]);
                }
            }
        };
        auto mask_tail = [&]() {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (k0 + kt * 32 + crow(r, g) >= Tk) s[qb][kt][r] = -INFINITY;
        };
        scores();
        // next tile's loads are issued behind the QK^T MFMAs and fly under the softmax and the PV MFMAs (issued in front of them the
        // compiler parks an s_waitcnt vmcnt(0) before the first MFMA and the whole load latency is exposed every tile)
        if (k0 + 128 <= Tk) {                 // next tile is an interior one: planned, unchecked loads
            load_planned<D>(kreg, Kb + (int64_t)(k0 + 64) * ldk, kplan);
            load_planned<D>(vreg, Vtb + (int64_t)(k0 + 64) * ldk, vplan);
        } else if (k0 + 64 < Tk) {
            load_rows<D>(kreg, Kb, ldk, k0 + 64, Tk);
            load_rows<D>(vreg, Vtb, ldk, k0 + 64, Tk);
        }
        // online softmax on the raw scores: p = exp2(s*sl2 - m*sl2) is one FMA + one v_exp per element; the
        // key mask only exists in the last (partial) tile, a wave-uniform branch
        if (k0 + 64 > Tk) {
            asm volatile("" ::: "memory");   // keeps this a real (wave-uniform) branch: if-converted it costs ~90 VALU on every tile
            mask_tail();
        }
        f16x8 pf[QB][4];
        if constexpr (PRE) {
            // the accumulator already holds log2-domain scores relative to m_run: p = exp2(s), no arithmetic in front of the exponential.  The reference
            // point moves on the first tile and afterwards only when a tile's maximum exceeds it by more than 2^LAZY -- a wave-uniform, rarely taken
            // branch that rescales O, rewrites q's three slots with the new -m_run and simply RE-RUNS the tile's QK^T MFMAs against them
            bool mv[QB];
            float mxv[QB];
            bool any_mv = false;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][kt][r]);
                mxv[qb] = fmaxf(mx, __shfl_xor(mx, 32, 64));
                mv[qb] = k0 == 0 || mxv[qb] > FD_ATTN_LAZY;
                any_mv = any_mv || mv[qb];
            }
            if (__any(any_mv)) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    const float delta = mv[qb] ? mxv[qb] : 0.f;
                    // the first tile always moves the reference point, possibly DOWN (all of its scores below -128: delta < -128, exp2(-delta) = inf and
                    // inf * 0 = NaN in the still-zero O / l): nothing has been accumulated before it, so the factor there is 1
                    const float alpha = k0 == 0 ? 1.f : __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int i = 0; i < NDV; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[qb][i][r] *= alpha;
                    if (!ONES) l_run[qb] *= alpha;
                    m_run[qb] += delta;
                    if (g == 1) {
                        f16 h0, h1, h2;
                        split3(-m_run[qb], h0, h1, h2);
                        qf[qb][NKS - 1][0] = h0; qf[qb][NKS - 1][1] = h1; qf[qb][NKS - 1][2] = h2;
                    }
                }
                scores();
                if (k0 + 64 > Tk) mask_tail();
            }
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float rs = 0.f;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(s[qb][kt][r]);
                        if (!ONES) rs += p;
                        pf[qb][kt * 2 + (r >> 3)][r & 7] = (f16)p;
                    }
                if (!ONES) {
                    rs += __shfl_xor(rs, 32, 64);
                    l_run[qb] += rs;
                }
            }
        } else
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][kt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // Lazy reference point (round 4): the running "max" only moves when a tile's maximum exceeds it by more than 2^LAZY in the
            // probability domain, so p <= 2^LAZY (exact in fp16, fp32 row sums) and after the first tiles the 32 v_mul of the O rescale
            // and its wave-wide vote almost never execute.  O / l and LSE = m + log l do not depend on the reference point.
            constexpr float LAZY = FD_ATTN_LAZY;
            const float m_new = (mx - m_run[qb]) * sl2 > LAZY ? mx : m_run[qb];          // reference point of the RAW scores (-inf at the start: first tile always moves it)
            const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * sl2);
            const float nm = -m_new * sl2;
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[qb][kt][r], sl2, nm));
                    if (!ONES) rs += p;
                    pf[qb][kt * 2 + (r >> 3)][r & 7] = (f16)p;
                }
            if (!ONES) {
                rs += __shfl_xor(rs, 32, 64);
                l_run[qb] = l_run[qb] * alpha + rs;
            }
            if (__any(m_new != m_run[qb])) {                   // rescale O only when some row's max moved
#pragma unroll
                for (int i = 0; i < NDV; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[qb][i][r] *= alpha;
            }
            m_run[qb] = m_new;
        }
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int i = 0; i < NDV; ++i) {
                const f16x8 vf = read_tr(Vts, VP, st * 16, i * 32, ql, g);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) oacc[qb][i] = mfma32(vf, pf[qb][st], oacc[qb][i]);
            }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        if (ONES) {                                    // row D of O^T: lane-half GL, register RL of tile D / 32 (crow), for this lane's query column
            constexpr int LOC = D % 32, GL = (LOC >> 2) & 1, RL = (LOC & 3) + 4 * (LOC >> 3);
            l_run[qb] = __shfl(oacc[qb][D / 32][RL], ql + 32 * GL, 64);
        }
        if (tvalid[qb]) {
            const float inv = 1.f / l_run[qb];
            f16* Op = O + ((int64_t)b * Tq + t[qb]) * C + h * D;
#pragma unroll
            for (int i = 0; i < NDV; ++i)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int dv = i * 32 + 8 * rq + 4 * g;
                    if (dv < D) {
                        f16x4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (f16)(oacc[qb][i][rq * 4 + j] * inv);
                        *(f16x4*)(Op + dv) = o;
                    }
                }
            if (LSE && g == 0) LSE[((int64_t)b * H + h) * Tq + t[qb]] = PRE ? (m_run[qb] + log2f(l_run[qb])) / LOG2E : m_run[qb] * scale + log2f(l_run[qb]) / LOG2E;
        }
    }
}

// ================================================================================== D = rowsum(dO * O)
__global__ void attn_bwd_prep_kernel(const f16* O, const f16* dO, float* Dd, int H, int T, int d, int64_t n) {
    // one thread per (b, t, h)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int h = (int)(i % H);
        const int64_t bt = i / H;
        const int t = (int)(bt % T);
        const int64_t b = bt / T;
        const f16* o = O + bt * H * d + h * d;
        const f16* g = dO + bt * H * d + h * d;
        float s = 0.f;
        for (int j = 0; j < d; j += 8) {
            const f16x8 a = *(const f16x8*)(o + j), c = *(const f16x8*)(g + j);
#pragma unroll
            for (int k = 0; k < 8; ++k) s += (float)a[k] * (float)c[k];
        }
        Dd[(b * H + h) * T + t] = s;
    }
}

// ================================================================================== dQ
// no transposed K in HBM: the dQ operand K^T comes from the row-major K tile through read_tr
template <int D, bool PRE = false>
__global__ __launch_bounds__(256, dq_waves(D)) void attn_bwd_dq_kernel(const f16* __restrict__ Q, const f16* __restrict__ K, const f16* __restrict__ V,
                                                          const f16* __restrict__ dO,
                                                          const float* __restrict__ LSE, float* __restrict__ Dd, f16* __restrict__ dQ,
                                                          const f16* __restrict__ O, int H, int Tq, int Tk, int Tkr, int kv_div,
                                                          float scale, int ldq, int ldkv, int lddq) {
    FD_WG_TRACE(8);
    constexpr int DK = (D + 15) / 16 * 16, DV = (D + 31) / 32 * 32, DKP = DK + FD_ATTN_BWD_PAD;
    constexpr int NKS = DK / 16, NDV = DV / 32;
    extern __shared__ __attribute__((aligned(16))) f16 smem[];
    f16* Ks = smem;                // [64][DKP]
    f16* Vs = Ks + 64 * DKP;       // [64][DKP]

    int b, h, qblk;
    attn_block_coords((Tq + 127) / 128, H, gridDim.x / (((Tq + 127) / 128) * H), b, h, qblk);
    const int q0 = qblk * 128;
    const int bk = b / kv_div;
    const int C = H * D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 31, g = lane >> 5;
    const int t = q0 + wave * 32 + ql;
    const bool tvalid = t < Tq;

    f16x8 qf[NKS], gf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int col = ks * 16 + g * 8;
        qf[ks] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        gf[ks] = qf[ks];
        if (tvalid && col < D) {
            qf[ks] = *(const f16x8*)(Q + ((int64_t)b * Tq + t) * ldq + h * D + col);
            gf[ks] = *(const f16x8*)(dO + ((int64_t)b * Tq + t) * C + h * D + col);
        }
    }
    static_assert(!PRE || pre_ok<D>(), "pre-scaled q: head dims with three spare contraction slots");
    const float lse2 = tvalid ? LSE[((int64_t)b * H + h) * Tq + t] * LOG2E : (PRE ? FD_PRE_MASKED : INFINITY);
    if (PRE && g == 1) {                             // -lse of this lane's query into columns D .. D + 2 of its (pre-scaled) q row
        f16 h0, h1, h2;
        split3(-lse2, h0, h1, h2);
        qf[NKS - 1][0] = h0; qf[NKS - 1][1] = h1; qf[NKS - 1][2] = h2;
    }
    float dd;
    if (O) {   // D = rowsum(dO * O) computed here (each lane holds half of its query's columns) and published for the dK/dV kernel
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int col = ks * 16 + g * 8;
            if (tvalid && col < D) {
                const f16x8 ov = *(const f16x8*)(O + ((int64_t)b * Tq + t) * C + h * D + col);
#pragma unroll
                for (int j = 0; j < 8; ++j) part += (float)ov[j] * (float)gf[ks][j];
            }
        }
        dd = part + __shfl_xor(part, 32, 64);
        if (tvalid && g == 0) Dd[((int64_t)b * H + h) * Tq + t] = dd;
    } else {
        dd = tvalid ? Dd[((int64_t)b * H + h) * Tq + t] : 0.f;
    }
    const float sl2 = scale * LOG2E;
    constexpr bool DFOLD = dfold<D>();
    if (DFOLD && g == 1) {                           // this lane's columns D .. D + 2 of its query's dO row
        f16 h0, h1, h2;
        split3_scaled(-dd, h0, h1, h2);
        gf[NKS - 1][0] = h0; gf[NKS - 1][1] = h1; gf[NKS - 1][2] = h2;
    }
    f32x16 acc[NDV];
#pragma unroll
    for (int i = 0; i < NDV; ++i) acc[i] = zero16();

    const f16* Kb = K + (int64_t)bk * Tkr * ldkv + h * D;
    const f16* Vb = V + (int64_t)bk * Tkr * ldkv + h * D;

    constexpr bool PF = D <= 80;      // register prefetch where the register file has room
    TileRegs<D> kreg, vreg;
    zero_row_pad<D, DKP>(Ks);
    zero_row_pad<D, DKP>(Vs);
    if (DFOLD || PRE) {
        __syncthreads();                               // behind zero_row_pad's writes of the same columns
        for (int c = threadIdx.x; c < 64 * 3; c += 256) {    // never overwritten: store_rows writes columns < D
            if (DFOLD) Vs[(c / 3) * DKP + D + c % 3] = c % 3 == 0 ? (f16)256.f : c % 3 == 1 ? (f16)1.f : (f16)(1.f / 256.f);      // the scales of split3_scaled
            if (PRE) Ks[(c / 3) * DKP + D + c % 3] = (f16)1.f;
        }
    }
    if (PF) {
        load_rows<D>(kreg, Kb, ldkv, 0, Tk);
        load_rows<D>(vreg, Vb, ldkv, 0, Tk);
    }
    for (int k0 = 0; k0 < Tk; k0 += 64) {
        __syncthreads();
        if (!PF) {
            load_rows<D>(kreg, Kb, ldkv, k0, Tk);
            load_rows<D>(vreg, Vb, ldkv, k0, Tk);
        }
        store_rows<D, DKP>(kreg, Ks);
        store_rows<D, DKP>(vreg, Vs);
        __syncthreads();
        f16x8 dsf[4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const f16x8 kf = *(const f16x8*)(Ks + (kt * 32 + ql) * DKP + ks * 16 + g * 8);
                const f16x8 vf = *(const f16x8*)(Vs + (kt * 32 + ql) * DKP + ks * 16 + g * 8);
                s = mfma32(kf, qf[ks], s);
                dp = mfma32(vf, gf[ks], dp);
            }
            // next tile's loads go out behind the first MFMA group (in front of it the compiler waits for them at once, see forward)
            if (kt == 0 && PF && k0 + 64 < Tk) {
                load_rows<D>(kreg, Kb, ldkv, k0 + 64, Tk);
                load_rows<D>(vreg, Vb, ldkv, k0 + 64, Tk);
            }
            if (k0 + 64 > Tk) {                  // the key mask only exists in the last (partial) tile: a wave-uniform branch
                asm volatile("" ::: "memory");   // keeps it a real branch (if-converted it is ~110 VALU -- as many as the softmax itself -- on every tile)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + kt * 32 + crow(r, g) >= Tk) s[r] = -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(PRE ? s[r] : fmaf(s[r], sl2, -lse2));
                dsf[kt * 2 + (r >> 3)][r & 7] = (f16)(p * (DFOLD ? dp[r] : dp[r] - dd));   // the softmax scale is applied once to the accumulator
            }
        }
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int i = 0; i < NDV; ++i) {
                const f16x8 kf = read_tr(Ks, DKP, st * 16, i * 32, ql, g);
                acc[i] = mfma32(kf, dsf[st], acc[i]);
            }
    }
    if (tvalid) {
        f16* P = dQ + ((int64_t)b * Tq + t) * lddq + h * D;
#pragma unroll
        for (int i = 0; i < NDV; ++i)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int dv = i * 32 + 8 * rq + 4 * g;
                if (dv < D) {
                    f16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (f16)(acc[i][rq * 4 + j] * scale);     // PRE: the launcher passes the true softmax scale here too (d/dq, not d/dq')
                    *(f16x4*)(P + dv) = o;
                }
            }
    }
}

// ================================================================================== dK, dV
// block = 128 keys (4 waves x 32), loops over 32-query tiles.  S[q][key] = Q.K^T with K,V rows in VGPRs.
// no transposed Q / dO in HBM: the dK / dV operands Q^T, dO^T come from the row-major tiles through read_tr
template <int D, bool ATOMIC, bool PRE = false>
__global__ __launch_bounds__(256, dkdv_waves(D)) void attn_bwd_dkdv_kernel(const f16* __restrict__ Q, const f16* __restrict__ K,
                                                            const f16* __restrict__ V, const f16* __restrict__ dO,
                                                            const float* __restrict__ LSE,
                                                            const float* __restrict__ Dd, void* __restrict__ dKo, void* __restrict__ dVo,
                                                            int H, int Tq, int Tk, int Tkr, int kv_div, float scale, int ldq, int ldkv, int lddkv,
                                                            int64_t slab) {
    FD_WG_TRACE(9);
    // ATOMIC + slab > 0: no atomics -- sample j of a K/V group stores its fp32 partial into slab j (slab = elements per [Bk*Tkr, lddkv] buffer);
    // the caller sums the kv_div slabs in a fixed order (fd_sum_slabs): bit-reproducible shared dK / dV
    constexpr int DK = (D + 15) / 16 * 16, DV = (D + 31) / 32 * 32, DKP = DK + FD_ATTN_BWD_PAD;
    constexpr int NKS = DK / 16, NDV = DV / 32;
    extern __shared__ __attribute__((aligned(16))) f16 smem[];
    f16* Qs = smem;                 // [64][DKP]
    f16* Gs = Qs + 64 * DKP;        // [64][DKP]  (dO rows)
    float* lse_s = (float*)(Gs + 64 * DKP + 64);  // [64]  (64 halfs of slack: read_tr runs past the last row)
    float* dd_s = lse_s + 64;                // [64]

    int b, h, kblk;
    attn_block_coords((Tk + 127) / 128, H, gridDim.x / (((Tk + 127) / 128) * H), b, h, kblk);
    const int k0 = kblk * 128;
    const int bk = b / kv_div;
    const int C = H * D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kl = lane & 31, g = lane >> 5;
    const int key = k0 + wave * 32 + kl;
    const bool kvalid = key < Tk;

    f16x8 kf[NKS], vf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int col = ks * 16 + g * 8;
        kf[ks] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        vf[ks] = kf[ks];
        if (kvalid && col < D) {
            kf[ks] = *(const f16x8*)(K + ((int64_t)bk * Tkr + key) * ldkv + h * D + col);
            vf[ks] = *(const f16x8*)(V + ((int64_t)bk * Tkr + key) * ldkv + h * D + col);
        }
    }
    constexpr bool DFOLD = dfold<D>();
    static_assert(!PRE || pre_ok<D>(), "pre-scaled q: head dims with three spare contraction slots");
    if (DFOLD && g == 1) {                                                                     // columns D .. D + 2 of this lane's V row: the scales of split3_scaled
        vf[NKS - 1][0] = (f16)256.f; vf[NKS - 1][1] = (f16)1.f; vf[NKS - 1][2] = (f16)(1.f / 256.f);
    }
    if (PRE && g == 1) kf[NKS - 1][0] = kf[NKS - 1][1] = kf[NKS - 1][2] = (f16)1.f;           // ... and of its K row: they meet -lse in q's
    f32x16 dk[NDV], dv[NDV];
#pragma unroll
    for (int i = 0; i < NDV; ++i) { dk[i] = zero16(); dv[i] = zero16(); }
    const float sl2 = scale * LOG2E;

    const f16* Qb = Q + (int64_t)b * Tq * ldq + h * D;
    const f16* Gb = dO + (int64_t)b * Tq * C + h * D;
    const float* Lb = LSE + ((int64_t)b * H + h) * Tq;
    const float* Db = Dd + ((int64_t)b * H + h) * Tq;

    constexpr bool PF = D <= 80;
    TileRegs<D> qreg, greg;
    zero_row_pad<D, DKP>(Qs);
    zero_row_pad<D, DKP>(Gs);
    if (PF) {
        load_rows<D>(qreg, Qb, ldq, 0, Tq);
        load_rows<D>(greg, Gb, C, 0, Tq);
    }
    for (int q0 = 0; q0 < Tq; q0 += 64) {
        __syncthreads();
        if (!PF) {
            load_rows<D>(qreg, Qb, ldq, q0, Tq);
            load_rows<D>(greg, Gb, C, q0, Tq);
        }
        store_rows<D, DKP>(qreg, Qs);
        store_rows<D, DKP>(greg, Gs);
        if (threadIdx.x < 64) {
            const int tq = q0 + threadIdx.x;
            const float lsev = tq < Tq ? Lb[tq] * LOG2E : (PRE ? FD_PRE_MASKED : INFINITY);
            if (PRE) {                                  // -lse of query row tq into columns D .. D + 2 of its (pre-scaled) q row
                f16 h0, h1, h2;
                split3(-lsev, h0, h1, h2);
                f16* qp = Qs + threadIdx.x * DKP + D;
                qp[0] = h0; qp[1] = h1; qp[2] = h2;
            } else lse_s[threadIdx.x] = lsev;
            const float ddv = tq < Tq ? Db[tq] : 0.f;
            if (DFOLD) {                                // -D of query row tq into columns D .. D + 2 of its dO row (store_rows writes columns < D only)
                f16 h0, h1, h2;
                split3_scaled(-ddv, h0, h1, h2);
                f16* gp = Gs + threadIdx.x * DKP + D;
                gp[0] = h0; gp[1] = h1; gp[2] = h2;
            } else dd_s[threadIdx.x] = ddv;
        }
        __syncthreads();
        f16x8 pf[4], dsf[4];
        // p and dS of keys past Tk are never zeroed: such a lane owns a key COLUMN of dK^T / dV^T that is never stored, and with its K / V
        // fragments zero everything it computes stays finite
        auto soft = [&](const f32x16& s, const f32x16& dp, int qt, int r) {
            const int qi = qt * 32 + crow(r, g);
            const float p = __builtin_amdgcn_exp2f(PRE ? s[r] : fmaf(s[r], sl2, -lse_s[qi]));
            pf[qt * 2 + (r >> 3)][r & 7] = (f16)p;
            dsf[qt * 2 + (r >> 3)][r & 7] = (f16)(p * (DFOLD ? dp[r] : dp[r] - dd_s[qi]));   // scale applied to dK at the end
        };
        auto dvdk = [&](int st, int i) {
            const f16x8 ga = read_tr(Gs, DKP, st * 16, i * 32, kl, g);
            const f16x8 qa = read_tr(Qs, DKP, st * 16, i * 32, kl, g);
            dv[i] = mfma32(ga, pf[st], dv[i]);
            dk[i] = mfma32(qa, dsf[st], dk[i]);
        };
#ifdef FD_DKDV_PIPE
        // measurement: both score tiles first, then the second tile's softmax interleaved with the first tile's dV / dK products
        f32x16 s2[2], dp2[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            s2[qt] = zero16(); dp2[qt] = zero16();
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const f16x8 qa = *(const f16x8*)(Qs + (qt * 32 + kl) * DKP + ks * 16 + g * 8);
                const f16x8 ga = *(const f16x8*)(Gs + (qt * 32 + kl) * DKP + ks * 16 + g * 8);
                s2[qt] = mfma32(qa, kf[ks], s2[qt]);
                dp2[qt] = mfma32(ga, vf[ks], dp2[qt]);
            }
        }
        if (PF && q0 + 64 < Tq) {
            load_rows<D>(qreg, Qb, ldq, q0 + 64, Tq);
            load_rows<D>(greg, Gb, C, q0 + 64, Tq);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) soft(s2[0], dp2[0], 0, r);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int r = 4 * c; r < 4 * c + 4; ++r) soft(s2[1], dp2[1], 1, r);
            if (NDV == 2) dvdk(c >> 1, c & 1);
            else if (c < 2) {
#pragma unroll
                for (int i = 0; i < NDV; ++i) dvdk(c, i);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int st = 2; st < 4; ++st)
#pragma unroll
            for (int i = 0; i < NDV; ++i) dvdk(st, i);
#else
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const f16x8 qa = *(const f16x8*)(Qs + (qt * 32 + kl) * DKP + ks * 16 + g * 8);
                const f16x8 ga = *(const f16x8*)(Gs + (qt * 32 + kl) * DKP + ks * 16 + g * 8);
                s = mfma32(qa, kf[ks], s);
                dp = mfma32(ga, vf[ks], dp);
            }
            if (qt == 0 && PF && q0 + 64 < Tq) {       // prefetch behind the first MFMA group (see forward)
                load_rows<D>(qreg, Qb, ldq, q0 + 64, Tq);
                load_rows<D>(greg, Gb, C, q0 + 64, Tq);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) soft(s, dp, qt, r);
        }
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int i = 0; i < NDV; ++i) dvdk(st, i);
#endif
    }
    if (kvalid) {
        const int64_t off = ((int64_t)bk * Tkr + key) * lddkv + h * D;
#pragma unroll
        for (int i = 0; i < NDV; ++i)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int d0 = i * 32 + 8 * rq + 4 * g;
                if (d0 < D) {
                    if (ATOMIC && slab > 0) {
                        const int64_t so = (int64_t)(b - bk * kv_div) * slab + off + d0;
                        *(f32x4*)((float*)dKo + so) = (f32x4){dk[i][rq * 4] * scale, dk[i][rq * 4 + 1] * scale, dk[i][rq * 4 + 2] * scale, dk[i][rq * 4 + 3] * scale};
                        *(f32x4*)((float*)dVo + so) = (f32x4){dv[i][rq * 4], dv[i][rq * 4 + 1], dv[i][rq * 4 + 2], dv[i][rq * 4 + 3]};
                    } else if (ATOMIC) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            atomicAdd((float*)dKo + off + d0 + j, dk[i][rq * 4 + j] * scale);
                            atomicAdd((float*)dVo + off + d0 + j, dv[i][rq * 4 + j]);
                        }
                    } else {
                        f16x4 a, c;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { a[j] = (f16)(dk[i][rq * 4 + j] * scale); c[j] = (f16)dv[i][rq * 4 + j]; }
                        *(f16x4*)((f16*)dKo + off + d0) = a;
                        *(f16x4*)((f16*)dVo + off + d0) = c;
                    }
                }
            }
    }
}

// ================================================================================== host side
// Query blocks per wave of the forward (1 or 2).  FD_ATTN_FWD_QB (bench-hooks build only) forces a value for A/B.
#ifdef FD_BENCH_HOOKS
#include <stdlib.h>
static inline const char* bench_env_attn(const char* n) { return getenv(n); }
#else
static inline const char* bench_env_attn(const char*) { return nullptr; }
#endif
#ifndef FD_ATTN_FWD_QB2
#define FD_ATTN_FWD_QB2 1        // shipped policy: two query blocks per wave for long sequences of the small head dims (0: always one)
#endif
static int fwd_qb(int d, int Tq, int Tk, int BH) {
    // measured, B16 H8 T4096 d40 (profiles/r04_attention_fwd_lazy_qb_ab.txt): 660 us (one block, eager reference point) -> 632 (lazy) -> 593 (lazy, two blocks);
    // T = 1024 / d = 80 and the 77-key cross attention do not gain (kept at one block)
    int qb = (FD_ATTN_FWD_QB2 && d <= 64 && Tq >= 2048 && Tk >= 2048 && (long)BH * ((Tq + 255) / 256) >= 128) ? 2 : 1;
    static const char* e = bench_env_attn("FD_ATTN_FWD_QB");
    if (e) qb = atoi(e) == 2 && d <= 64 ? 2 : 1;
    return qb;
}

template <int D> static constexpr size_t fwd_lds() {
    constexpr int DKP = (D + 15) / 16 * 16 + 8, DV = (D + 31) / 32 * 32;
    return (size_t)(64 * DKP + 64 * tr_stride(DV)) * 2;
}
template <int D> static constexpr size_t dq_lds() {
    constexpr int DKP = (D + 15) / 16 * 16 + FD_ATTN_BWD_PAD;
    return (size_t)(2 * 64 * DKP + 64) * 2;      // + slack for the transpose reads that run past the last row
}
template <int D> static constexpr size_t dkdv_lds() {
    constexpr int DKP = (D + 15) / 16 * 16 + FD_ATTN_BWD_PAD;
    return (size_t)(2 * 64 * DKP + 64) * 2 + 512;   // + the same slack + lse_s, dd_s
}

#define FD_DISPATCH_D(d, CALL)                                                       \
    switch (d) {                                                                     \
        case 16: { CALL(16); break; }                                                \
        case 32: { CALL(32); break; }                                                \
        case 40: { CALL(40); break; }                                                \
        case 64: { CALL(64); break; }                                                \
        case 80: { CALL(80); break; }                                                \
        case 128: { CALL(128); break; }                                              \
        case 160: { CALL(160); break; }                                              \
        default: fd_set_error("attention: unsupported head dim %d", d); return FD_ERR_ARG; \
    }

// raise the dynamic-LDS cap once per kernel instantiation (each macro expansion has its own static)
#define ALLOW_LDS(kern, bytes)                                                                                   \
    {                                                                                                            \
        static bool once = false;                                                                                \
        if (!once) {                                                                                             \
            (void)hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            once = true;                                                                                         \
        }                                                                                                        \
    }

extern "C" int fd_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Tq, int Tk, int Tkr,
                           int d, int kv_div, float scale, int ldq, int ldk, void* stream) {
    if (ldq <= 0) ldq = H * d;
    if (ldk <= 0) ldk = H * d;
    FD_REQUIRE((ldq & 7) == 0 && (ldk & 7) == 0, "fd_attn_fwd: row strides must be multiples of 8");
    FD_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0 && Tkr >= Tk && kv_div >= 1, "fd_attn_fwd: bad shape");
    // 64 queries per wave (QB = 2) for the long self-attention sequences of the small head dims: halves the LDS fragment traffic and the K / V
    // staging per query (see attn_fwd_kernel).  Short sequences keep QB = 1 (more workgroups, three waves per SIMD).
    const int qb = fwd_qb(d, Tq, Tk, B * H);
    dim3 grid(((Tq + 128 * qb - 1) / (128 * qb)) * H * B);
    // scale < 0: q arrives multiplied by |scale| * log2(e) (fd_gemm_desc.colscale in the projection) -- see "pre-scaled q" above
    const bool pre = scale < 0.f;
    scale = fabsf(scale);
    FD_REQUIRE(!pre || d % 16 == 8, "fd_attn_fwd: pre-scaled q (negative scale) needs d %% 16 == 8");
#define LAUNCH_F(DD, QBV, PREV)                                                                                                \
    {                                                                                                                          \
        ALLOW_LDS((attn_fwd_kernel<DD, QBV, PREV>), (fwd_lds<DD>()));                                                          \
        hipLaunchKernelGGL((attn_fwd_kernel<DD, QBV, PREV>), grid, dim3(256), (fwd_lds<DD>()), (hipStream_t)stream,            \
                           (const f16*)q, (const f16*)k, (const f16*)v, (f16*)o, lse, H, Tq, Tk, Tkr, kv_div, scale, ldq, ldk); \
    }
#define CALL(DD)                                                             \
    if constexpr (pre_ok<DD>()) {                                            \
        if (pre) {                                                           \
            if (qb == 2) LAUNCH_F(DD, (DD <= 64 ? 2 : 1), true)              \
            else LAUNCH_F(DD, 1, true)                                       \
            break;                                                           \
        }                                                                    \
    }                                                                        \
    if (qb == 2 && DD <= 64) LAUNCH_F(DD, (DD <= 64 ? 2 : 1), false)         \
    else LAUNCH_F(DD, 1, false)
    FD_DISPATCH_D(d, CALL)
#undef CALL
#undef LAUNCH_F
    return fd_check_launch("fd_attn_fwd");
}

extern "C" int fd_attn_bwd_prep(const void* o, const void* d_o, float* D, int B, int H, int T, int d, void* stream) {
    FD_REQUIRE((d & 7) == 0, "fd_attn_bwd_prep: d %% 8");
    const int64_t n = (int64_t)B * T * H;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)o, (const f16*)d_o, D, H, T, d, n);
    return fd_check_launch("fd_attn_bwd_prep");
}

extern "C" int fd_attn_bwd_dq(const void* q, const void* k, const void* v, const void* d_o, const float* lse, float* D,
                              const void* o, void* dq, int B, int H, int Tq, int Tk, int Tkr, int d, int kv_div, float scale,
                              int ldq, int ldkv, int lddq, void* stream) {
    if (ldq <= 0) ldq = H * d;
    if (ldkv <= 0) ldkv = H * d;
    if (lddq <= 0) lddq = H * d;
    FD_REQUIRE((ldq & 7) == 0 && (ldkv & 7) == 0 && (lddq & 3) == 0, "fd_attn_bwd_dq: row strides");
    FD_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0 && Tkr >= Tk && kv_div >= 1, "fd_attn_bwd_dq: bad shape");
    dim3 grid(((Tq + 127) / 128) * H * B);
    const bool pre = scale < 0.f;                    // q arrives multiplied by |scale| * log2(e): see "pre-scaled q"
    scale = fabsf(scale);
    FD_REQUIRE(!pre || d % 16 == 8, "fd_attn_bwd_dq: pre-scaled q (negative scale) needs d %% 16 == 8");
#define LAUNCH_Q(DD, PREV)                                                                                                             \
    {                                                                                                                                  \
        ALLOW_LDS((attn_bwd_dq_kernel<DD, PREV>), (dq_lds<DD>()));                                                                     \
        hipLaunchKernelGGL((attn_bwd_dq_kernel<DD, PREV>), grid, dim3(256), (dq_lds<DD>()), (hipStream_t)stream, (const f16*)q,        \
                           (const f16*)k, (const f16*)v, (const f16*)d_o, lse, D, (f16*)dq, (const f16*)o, H, Tq, Tk, Tkr, kv_div,     \
                           scale, ldq, ldkv, lddq);                                                                                    \
    }
#define CALL(DD)                      \
    if constexpr (pre_ok<DD>()) {     \
        if (pre) {                    \
            LAUNCH_Q(DD, true)        \
            break;                    \
        }                             \
    }                                 \
    LAUNCH_Q(DD, false)
    FD_DISPATCH_D(d, CALL)
#undef CALL
#undef LAUNCH_Q
    return fd_check_launch("fd_attn_bwd_dq");
}

extern "C" int fd_attn_bwd_dkdv(const void* q, const void* k, const void* v, const void* d_o,
                                const float* lse, const float* D, void* dk, void* dv, int B, int H, int Tq, int Tk, int Tkr, int d,
                                int kv_div, float scale, int ldq, int ldkv, int lddkv, int accumulate, void* stream) {
    if (ldq <= 0) ldq = H * d;
    if (ldkv <= 0) ldkv = H * d;
    if (lddkv <= 0) lddkv = H * d;
    FD_REQUIRE((ldq & 7) == 0 && (ldkv & 7) == 0 && (lddkv & 3) == 0, "fd_attn_bwd_dkdv: row strides");
    FD_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0 && Tkr >= Tk && kv_div >= 1, "fd_attn_bwd_dkdv: bad shape");
    dim3 grid(((Tk + 127) / 128) * H * B);
    // accumulate == 2: fp32 per-sample slabs instead of atomics (dk, dv: [kv_div][Bk*Tkr][lddkv] fp32, every element written)
    const int64_t slab = accumulate == 2 ? (int64_t)(B / kv_div) * Tkr * lddkv : 0;
    FD_REQUIRE(accumulate != 2 || (B % kv_div) == 0, "fd_attn_bwd_dkdv: B must be a multiple of kv_div");
    // scale < 0: q arrives multiplied by |scale| * log2(e) ("pre-scaled q"): dK = scale * dS^T . q = dS^T . q' / log2(e)
    const bool pre = scale < 0.f;
    scale = pre ? 1.f / LOG2E : scale;
    FD_REQUIRE(!pre || d % 16 == 8, "fd_attn_bwd_dkdv: pre-scaled q (negative scale) needs d %% 16 == 8");
#define LAUNCH(DD, AT, PREV)                                                                                                           \
    {                                                                                                                                  \
        ALLOW_LDS((attn_bwd_dkdv_kernel<DD, AT, PREV>), (dkdv_lds<DD>()));                                                             \
        hipLaunchKernelGGL((attn_bwd_dkdv_kernel<DD, AT, PREV>), grid, dim3(256), (dkdv_lds<DD>()), (hipStream_t)stream,               \
                           (const f16*)q, (const f16*)k, (const f16*)v, (const f16*)d_o, lse, D, dk,                                   \
                           dv, H, Tq, Tk, Tkr, kv_div, scale, ldq, ldkv, lddkv, slab);                                                 \
    }
#define CALL(DD)                                                     \
    if constexpr (pre_ok<DD>()) {                                    \
        if (pre) {                                                   \
            if (kv_div > 1 || accumulate) LAUNCH(DD, true, true) else LAUNCH(DD, false, true) \
            break;                                                   \
        }                                                            \
    }                                                                \
    if (kv_div > 1 || accumulate) LAUNCH(DD, true, false) else LAUNCH(DD, false, false)
    FD_DISPATCH_D(d, CALL)
#undef CALL
#undef LAUNCH
    return fd_check_launch("fd_attn_bwd_dkdv");
}

FD_WGT_SETTER(attn)
