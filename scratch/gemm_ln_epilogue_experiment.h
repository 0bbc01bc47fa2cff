// The LayerNorm-as-second-output epilogue of gemm_big_kernel<BM, 320, 4, 4, 5> as it stood when it left the product in round 5:
// built and parity-tested in round 4, never a win inside the step (profiles/r04_layernorm_epilogue.txt: isolated 58 -> 50 us per pair, in situ 73 us against
// ~33 + 29, whole step 1361-1381 vs 1361-1372 ms).  It was wired as: fd_gemm_desc fields ln_out / ld_ln / ln_gamma / ln_beta / ln_stats / ln_eps (ABI 3),
// fd_gemm_ln_ok(), template variant CV = 5, ops.LN_EPILOGUE (FD_LN_EPILOGUE=1), tests test_gemm_layernorm_epilogue* (git history: round 4).

// ---- LayerNorm of the output row as a second output (fd_gemm_desc.ln_out; VERDICT r3 item 5 / row x2: "LayerNorm into its GEMM", producer side).
// Only where the workgroup tile holds WHOLE rows (N == BN == 320: proj_in, attn1.to_out, attn2.to_out of the 64^2 level): a row's 320 columns sit in
// the WGN = 4 waves of one wave row, 80 columns each.  Per pass of 32 staged rows: a wave stores its block of C exactly as gemm_epilogue_lds does
// (bias in the staging, residual on the way out) and keeps the values it stored in registers; row sums and sums of squares of its 80 columns (v_dot2c
// pair accumulators, lane -> wave-private LDS -> one lane per row, fixed order) go to a workgroup-shared table, one workgroup barrier, every lane adds the
// four wave partials of its rows (wn ascending), normalises its values and writes the second output.  The standalone LayerNorm pass -- a read and a
// write of the tensor -- becomes a write.  Statistics: mean = S / N, var = Q / N - mean^2 in fp32 (|x| <= a few tens, N = 320: the cancellation
// error stays below 1e-4 of the fp16 output's own rounding), rstd = rsqrt(var + eps), mean / rstd saved for fd_layernorm_bwd.
template <int TM, int TN, int TMC, int WGM, int WGN>
__device__ __forceinline__ void gemm_epilogue_ln(const fd_gemm_desc& p, f32x4 (&acc)[TM][TN], f16* wave_lds, float* xrow, int wm, int wn,
                                                 int mbase, int nbase, int lane) {
    constexpr int WTN = TN * 16, WTMC = TMC * 16, LDW = WTN + 4;
    constexpr int CPR = WTN / 8, RPI = 64 / CPR, NIT = (WTMC + RPI - 1) / RPI, NROW = NIT * RPI;
    static_assert(WTN == 80 && TM % TMC == 0 && NROW <= 64, "LayerNorm epilogue geometry");
    const int l15 = lane & 15, lg = lane >> 4;
    const int cr = lane / CPR, chunk = lane % CPR, cc = chunk * 8;
    const f16* R = (const f16*)p.residual;
    const f16x2 ones = {(f16)1.f, (f16)1.f};
    const float inv_n = 1.f / (float)p.N;
    float2* xr = (float2*)xrow;                      // [2][WGM][NROW][WGN] wave partials of the workgroup (double-buffered by pass parity)
#pragma unroll
    for (int c0 = 0; c0 < TM; c0 += TMC) {
        const int par = (c0 / TMC) & 1;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = nbase + j * 16 + lg * 4;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bv = *(const f32x4*)(p.bias + n);
#pragma unroll
            for (int ii = 0; ii < TMC; ++ii) {
                const f32x4 v = acc[c0 + ii][j] + bv;
                *(f16x4*)(wave_lds + (ii * 16 + l15) * LDW + j * 16 + lg * 4) = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // pass A: store the block of C; the stored values go back into the lane's own staging slot (the normalisation re-reads them: held in registers
        // across the workgroup barrier they pushed the 128-register kernel into ~280 scratch accesses per lane, 117 us for a 30 us GEMM; so did
        // requesting the residual rows of all iterations up front, 71 us); row sums by a segmented shuffle over the row's 10 lanes (fixed tree), lane 0
        // of the row files the wave's partial
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = it * RPI + cr;
            const int m = mbase + c0 * 16 + row, n = nbase + cc;
            const bool ok = cr < RPI && row < WTMC && m < p.M;
            float sq0 = 0.f, sq1 = 0.f;
            if (ok) {
                const f16x4 lo = *(const f16x4*)(wave_lds + row * LDW + cc);
                const f16x4 hi = *(const f16x4*)(wave_lds + row * LDW + cc + 4);
                f16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (R) {
                    const f16x8 rv = *(const f16x8*)(R + (int64_t)m * p.ldr + n);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (f16)((float)v[k] + (float)rv[k]);
                    *(f16x4*)(wave_lds + row * LDW + cc) = (f16x4){v[0], v[1], v[2], v[3]};
                    *(f16x4*)(wave_lds + row * LDW + cc + 4) = (f16x4){v[4], v[5], v[6], v[7]};
                }
                *(f16x8*)((f16*)p.C + (int64_t)m * p.ldc + n) = v;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f16x2 pr = {v[2 * k], v[2 * k + 1]};
                    sq0 = FD_DOT2(pr, ones, sq0);
                    sq1 = FD_DOT2(pr, pr, sq1);
                }
            }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const float t0 = __shfl_down(sq0, off, 64), t1 = __shfl_down(sq1, off, 64);
                if (chunk + off < CPR) { sq0 += t0; sq1 += t1; }
            }
            if (chunk == 0 && cr < RPI) xr[((par * WGM + wm) * NROW + row) * WGN + wn] = make_float2(sq0, sq1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // LDS only: the global stores of C stay in flight across the barrier
        // pass B: every lane adds the four wave partials of its rows (wn ascending), normalises the values it stored and writes the second output
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = it * RPI + cr;
            const int m = mbase + c0 * 16 + row, n = nbase + cc;
            const bool ok = cr < RPI && row < WTMC && m < p.M;
            if (ok) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int w = 0; w < WGN; ++w) {
                    const float2 t = xr[((par * WGM + wm) * NROW + row) * WGN + w];
                    s += t.x;
                    q += t.y;
                }
                const float mean = s * inv_n;
                const float rstd = rsqrtf(fmaxf(q * inv_n - mean * mean, 0.f) + p.ln_eps);
                const float mr = mean * rstd;
                const f16x4 lo = *(const f16x4*)(wave_lds + row * LDW + cc);
                const f16x4 hi = *(const f16x4*)(wave_lds + row * LDW + cc + 4);
                const f32x4 g0 = *(const f32x4*)(p.ln_gamma + n), g1 = *(const f32x4*)(p.ln_gamma + n + 4);
                const f32x4 b0 = *(const f32x4*)(p.ln_beta + n), b1 = *(const f32x4*)(p.ln_beta + n + 4);
                f16x8 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o[k] = (f16)(((float)lo[k] * rstd - mr) * g0[k] + b0[k]);
                    o[4 + k] = (f16)(((float)hi[k] * rstd - mr) * g1[k] + b1[k]);
                }
                *(f16x8*)((f16*)p.ln_out + (int64_t)m * p.ld_ln + n) = o;
                if (wn == 0 && chunk == 0 && p.ln_stats) *(float2*)(p.ln_stats + (int64_t)m * 2) = make_float2(mean, rstd);
            }
        }
        if (c0 + TMC < TM) {   // the next pass overwrites the staging rows: this wave's reads must have returned
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

