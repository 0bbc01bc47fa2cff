// REJECTED experiment of round 2 (kept for the record; included by scratch/attn_fwd_experiment.hip).
// Direct-to-LDS, three-buffer, one-barrier-per-tile K / V pipeline for the attention forward (global_load_lds_dwordx4 with an XOR-swizzled
// V^T image), NW waves per workgroup.  Bit-identical to the shipped register-staged kernel (max |dO| = 0, max |dLSE| = 0 on B = 16, H = 8,
// T = 4096, d = 40) and NOT faster: 658-667 us vs 624-635 us (NW = 4), 685 vs 681 (NW = 8 / 16, a slower box) -- although the ablation of the
// staged kernel attributes 27 % of its time to staging.  Ablating THIS kernel: without the in-loop loads 554 us, without the barrier 651 us:
// the cost sits in delivering 10 KB per 64 keys per workgroup (2.7 GB per call, ~4.3 TB/s out of L2 / Infinity Cache into LDS), not in how it
// is staged, and halving the re-reads with 256-query workgroups did not move it either (685 vs 719).  119 registers (4 waves / SIMD).
// With a raw s_barrier instead of __syncthreads() (whose fence drains vmcnt(0), i.e. the tile that should stay in flight): 676 us (NW = 4),
// 659 us (NW = 8) against 637 us for the register-staged kernel on the same box -- still behind.
// 16 bytes per lane from global memory straight into LDS: lane l lands at (wave-uniform) lds_dst + 16 * l
__device__ __forceinline__ void glds16(const f16* src, f16* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int D> static constexpr size_t fwd_glds_lds() { return (size_t)(3 * (64 * D + D * 64) + (((D + 31) / 32 * 32) - D) * 64 + 8) * 2; }

// ================================================================================== forward, direct-to-LDS K / V pipeline
// Same arithmetic as attn_fwd_kernel, different operand delivery for the long self-attention shapes (Tk a multiple of 64, >= 3 tiles):
// the K and V^T tiles of 64 keys go from global memory straight into LDS (global_load_lds_dwordx4, no staging registers, no ds_write),
// three buffers deep, ONE barrier per tile:
//     iteration j:  s_waitcnt vmcnt(loads of tile j+1 still in flight) ; barrier ; issue tile j+2 -> buffer (j+2)%3 ; compute tile j
// The ablation of the register-staged kernel (profiles/r02_attn_fwd_d40_ablation.txt) put 27 % of its time into exactly that staging path.
// LDS images: K tile [64][D] unpadded (fragment reads at row stride D halfs: conflict-free for D = 40 / 80 / 160 over the b128 lane
// groups; for D = 40 the third 16-wide k-step reads 8 halfs into the next row, multiplied by the zero-padded q columns); V^T tile [D][64]
// with the 16-byte chunk index XOR-ed with (row & 7) on the load side so that the 8-byte operand reads of 32 consecutive rows spread over
// the banks (2-way at worst, as the padded layout of the staged kernel).  Rows D..DV-1 of the V^T operand are whatever follows in LDS
// (finite or not): they only feed output rows that are never stored.
template <int D, int NW>
__global__ __launch_bounds__(64 * NW, fwd_waves(D)) void attn_fwd_glds_kernel(const f16* __restrict__ Q, const f16* __restrict__ K,
                                                                          const f16* __restrict__ Vt, f16* __restrict__ O,
                                                                          float* __restrict__ LSE, int H, int Tq, int Tk, int Tkp, int Tkr,
                                                                          int kv_div, float scale, int ldq, int ldk) {
    constexpr int DK = (D + 15) / 16 * 16, DV = (D + 31) / 32 * 32;
    constexpr int NKS = DK / 16, NDV = DV / 32, CH = D / 8;
    constexpr int KT = 64 * D, VT = D * 64, BUF = KT + VT;          // halfs per tile / per buffer
    constexpr int NI = D / 4, NIK = NI / 2;                         // wave-instructions per tile pair (K + V^T) / per K tile
    extern __shared__ __attribute__((aligned(16))) f16 smem[];     // 3 buffers + (DV - D) * 64 + 8 halfs of slack behind the last one

    constexpr int QB = 32 * NW;          // query rows per workgroup: NW = 8 halves the K / V re-reads of NW = 4
    int b, h, qblk;
    attn_block_coords((Tq + QB - 1) / QB, H, gridDim.x / (((Tq + QB - 1) / QB) * H), b, h, qblk);
    const int q0 = qblk * QB;
    const int bk = b / kv_div;
    const int C = H * D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 31, g = lane >> 5;
    const int t = q0 + wave * 32 + ql;
    const bool tvalid = t < Tq;

    f16x8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int col = ks * 16 + g * 8;
        qf[ks] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (tvalid && col < D) qf[ks] = *(const f16x8*)(Q + ((int64_t)b * Tq + t) * ldq + h * D + col);
    }
    f32x16 oacc[NDV];
#pragma unroll
    for (int i = 0; i < NDV; ++i) oacc[i] = zero16();
    float m_run = -INFINITY, l_run = 0.f;
    const float sl2 = scale * LOG2E;
    const f16* Kb = K + (int64_t)bk * Tkr * ldk + h * D;
    const f16* Vtb = Vt + ((int64_t)bk * C + h * D) * Tkp;

    // this wave's share of a tile pair: wave-instructions i = wave, wave + 4, ...  (per-lane source offsets are tile-invariant)
    constexpr int NMAX = (NI + NW - 1) / NW;
    unsigned soff[NMAX];                 // byte offset of the lane's 16 bytes relative to the tile's K (or V^T) base
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
        const int i = wave + NW * n;
        if (i < NIK) {
            const int p = 64 * i + lane, row = p / CH, c = p - row * CH;
            soff[n] = (unsigned)((row * ldk + c * 8) * 2);
        } else {
            const int p = 64 * (i - NIK) + lane, row = p >> 3, c = (p & 7) ^ (row & 7);
            soff[n] = (unsigned)((row * Tkp + c * 8) * 2);
        }
    }
    const int nmine = (NI - wave + NW - 1) / NW;  // 2 or 3 at D = 40, NW = 4
    auto issue = [&](int tile, int buf) {
        f16* base = smem + buf * BUF;
        const char* kg = (const char*)(Kb + (int64_t)tile * 64 * ldk);
        const char* vg = (const char*)(Vtb + tile * 64);
#pragma unroll
        for (int n = 0; n < NMAX; ++n) {
            const int i = wave + NW * n;
            if (i < NI) {
                if (i < NIK) glds16((const f16*)(kg + soff[n]), base + 64 * i * 8);
                else glds16((const f16*)(vg + soff[n]), base + KT + 64 * (i - NIK) * 8);
            }
        }
    };
    for (int c = threadIdx.x; c < (DV - D) * 64 + 8; c += 64 * NW) smem[3 * BUF + c] = (f16)0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // Q fragments landed: the counted waits below see only tile loads
    const int nt = Tk / 64;
    issue(0, 0);
    if (nt > 1) issue(1, 1);
    for (int j = 0; j < nt; ++j) {
        if (j + 1 < nt) {
            if (nmine == NMAX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NMAX) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NMAX - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#ifndef FD_GLDS_ABL_NOBAR
#ifdef FD_GLDS_SYNCTHREADS
        __syncthreads();                                   // its fence waits vmcnt(0): drains the tile that should stay in flight
#else
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // raw barrier: the counted vmcnt above is the only VM wait
        asm volatile("" ::: "memory");
#endif
#endif
#ifndef FD_GLDS_ABL_NOLOAD
        if (j + 2 < nt) issue(j + 2, (j + 2) % 3);
#endif
        const f16* Ks = smem + (j % 3) * BUF;
        const f16* Vts = Ks + KT;
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            s[kt] = zero16();
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const f16x8 kf = *(const f16x8*)(Ks + (kt * 32 + ql) * D + ks * 16 + g * 8);
                s[kt] = mfma32(kf, qf[ks], s[kt]);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2);
        const float nm = -m_new * sl2;
        float rs = 0.f;
        f16x8 pf[4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], sl2, nm));
                rs += p;
                pf[kt * 2 + (r >> 3)][r & 7] = (f16)p;
            }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        if (__any(m_new != m_run)) {
#pragma unroll
            for (int i = 0; i < NDV; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        }
        m_run = m_new;
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int i = 0; i < NDV; ++i) {
                const int row = i * 32 + ql;
                const f16* rp = Vts + row * 64 + g * 4;
                const f16x4 lo = *(const f16x4*)(rp + (((2 * st) ^ (row & 7)) << 3));
                const f16x4 hi = *(const f16x4*)(rp + (((2 * st + 1) ^ (row & 7)) << 3));
                const f16x8 vf = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                oacc[i] = mfma32(vf, pf[st], oacc[i]);
            }
    }
    if (tvalid) {
        const float inv = 1.f / l_run;
        f16* Op = O + ((int64_t)b * Tq + t) * C + h * D;
#pragma unroll
        for (int i = 0; i < NDV; ++i)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int dv = i * 32 + 8 * rq + 4 * g;
                if (dv < D) {
                    f16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (f16)(oacc[i][rq * 4 + j] * inv);
                    *(f16x4*)(Op + dv) = o;
                }
            }
        if (LSE && g == 0) LSE[((int64_t)b * H + h) * Tq + t] = m_run * scale + log2f(l_run) / LOG2E;
    }
}

