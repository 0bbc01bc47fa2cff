/* libfairdiff_hip.so -- C-ABI of the MI355X (gfx950) fairness-finetuning hot path.
 *
 * The reference (sail-sg/finetune-fair-diffusion) has no FFI/plugin interface: its hot
 * path is Python calling un-vendored diffusers/transformers/torchvision modules
 * (SURVEY.md section 8b).  Each entry point below names the reference interface whose
 * arithmetic it replaces (file:line into /root/reference) so a maintainer can bind it
 * from the reference side with ctypes.
 *
 * Conventions: every pointer is a DEVICE pointer unless stated; `stream` is a
 * hipStream_t passed as void*; the caller owns all buffers (kernels never allocate);
 * functions return FD_OK (0) or a negative FD_ERR_* code and are re-entrant (no global
 * mutable state except the thread-local last-error string).  Activations are fp16
 * ("wd" in SURVEY 8a), channels-last: an image tensor is [B, H, W, C] == a token matrix
 * [B*H*W, C]; LoRA/optimizer state and all reductions are fp32.
 */
#ifndef FAIRDIFF_HIP_H
#define FAIRDIFF_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FD_OK 0
#define FD_ERR_ARG (-1)
#define FD_ERR_LAUNCH (-2)
/* ABI revision of this header == fd_version() of a matching library.  3: every descriptor struct starts with ``struct_size`` (below);
 * 4: fd_gemm_desc lost the ln_* fields and fd_gemm_ln_ok(), the e4m3 attention entry points and fd_phase_shuffle are gone (round 5: scratch/),
 * the attention entry points take V / K / Q / dO as written by the projections only (no transposed-copy operands);
 * new: fd_cross_attn_block. */
#define FD_ABI_VERSION 4

enum { FD_ACT_NONE = 0, FD_ACT_SILU = 1, FD_ACT_QUICK_GELU = 2, FD_ACT_GELU = 3, FD_ACT_RELU = 4,
       FD_ACT_HARDSWISH = 5, FD_ACT_HARDSIGMOID = 6,
       /* fd_gemm only: B rows (and bias) interleaved (value_c, gate_c); C [M, N/2] = value * gelu_erf(gate), both rounded to fp16 first
          (== fd_geglu_fwd on the unfused projection, diffusers GEGLU); ``residual`` is then an optional second
          OUTPUT [M, N] (ldr) receiving the pre-gate projection in the same interleaved column order (for fd_geglu_bwd_interleaved) */
       FD_ACT_GEGLU = 7 };
enum { FD_OUT_F16 = 0, FD_OUT_F32 = 1 };
enum { FD_CONV_NORMAL = 0, FD_CONV_STRIDE2 = 1, FD_CONV_UP2 = 2, FD_CONV_TRANS2 = 3,
       /* conv3x3(nearest-up2(x)) (Upsample2D, diffusers resnet.py) as four 2x2-tap phase problems over the low-res input: A = x [Bn,H,W,Cin],
        * M = Bn*H*W per phase, K = 4*Cin (k = (dy*2+dx)*Cin + c), B = [4][N][K] pre-summed phase weights, C = [4][M][N] phase-major
        * (phase = py*2+px is output pixel (2y+py, 2x+px)); 4/9 of the multiply-adds.  Big-tile kernels only.                          */
       FD_CONV_UP2P = 4,
       /* its input gradient: A = dOut [Bn,H,W,Cin] at the HIGH resolution (H = 2*Ho), M = Bn*Ho*Wo, K = 16*Cin
        * (k = (((py*2+px)*2+dy)*2+dx)*Cin + c), B = [N][K].                                                                              */
       FD_CONV_UP2P_BWD = 5,
       /* FD_CONV_UP2P writing its four phases straight into the channels-last result: C = [Bn, 2H, 2W, N], the row of phase (py, px) and low-res pixel
        * (b, y, x) being ((b*2H + 2y+py)*2W + 2x+px): the form the product uses (the phase-major one needs a second pass to interleave).  gn_stats is allowed here: chunk (phase, m / 32) of the low-res rows,
        * i.e. slot phase * (M / 32) + m / 32 (M % 32 == 0); fd_groupnorm_fwd_stats takes that layout through ``per`` (chunks per image and phase). */
       FD_CONV_UP2PI = 6 };

const char* fd_last_error(void);
int fd_version(void);               /* == FD_ABI_VERSION of the header the library was built from */
/* "fp16" (libfairdiff_hip.so: the reference's mixed_precision fp16, configs 1-4) or "bf16" (libfairdiff_hip_bf16.so, built from the same
 * sources with -DFD_BF16: BASELINE configs[4]).  Every "fp16" in the prototypes below means this 16-bit working dtype. */
const char* fd_working_dtype(void);
/* "packed_fp32=off" when every translation unit was compiled without packed-fp32 VALU code (-fno-slp-vectorize, -packed-fp32-ops): the
 * loader refuses a library built otherwise -- such sequences returned wrong lanes under multi-stream SIMD sharing on gfx950. */
const char* fd_build_info(void);

/* ---- MFMA GEMM  C[M,N] = act(alpha * (A[M,K] . B[N,K]^T + A2[M,K2] . B2[N,K2]^T) + bias + rowbias) + residual
 * Replaces torch.nn.Linear / 1x1 Conv2d inside diffusers Attention/FeedForward/Transformer2DModel/
 * ResnetBlock2D (called through unet(...) at exp-1-debias-gender/1-main-debias.py:1046-1050,1118-1122)
 * and the LoRA rank update of LoRAAttnProcessor (injected at :798-818): the second (A2,B2) slab
 * carries t = x.down^T and the up matrix so the rank update is accumulated in the same MFMA tile.
 * The A operand may instead be gathered as an implicit-GEMM 3x3 convolution (conv != 0):
 * A is then a channels-last image [B,H,W,Cin] and K = 9*Cin with k = (ky*3+kx)*Cin + ci.        */
typedef struct fd_gemm_desc {
    /* FIRST field of every descriptor struct: sizeof(the struct) as the CALLER compiled it.  The entry points compare it with their own sizeof and
     * fail with FD_ERR_ARG (fd_last_error() names both sizes) on a mismatch, so a binding written against an older header -- the struct has grown
     * at its tail several times -- is refused instead of having fields read past its allocation. */
    int32_t struct_size;
    const void* A;  int64_t lda;        /* fp16 [M,K] row-major (or image when conv) */
    const void* B;  int64_t ldb;        /* fp16 [N,K] row-major */
    const void* A2; int64_t lda2;       /* optional second K-slab (LoRA / channel-concat) */
    const void* B2; int64_t ldb2;
    void* C;        int64_t ldc;        /* fp16 or fp32 [M,N] */
    const float* bias;                  /* fp32 [N] or NULL */
    const void* rowbias; int64_t ld_rowbias; int32_t rows_per_batch; /* fp16 [M/rows_per_batch, N] or NULL */
    const void* residual; int64_t ldr;  /* fp16 [M,N] or NULL */
    float alpha;
    int32_t M, N, K, K2;
    int32_t act, out_dtype;
    int32_t batch; int64_t sA, sB, sC, sR; /* strided batching (batch>=1), element strides */
    /* implicit-GEMM convolution gather (conv: 0 = dense A, 1 = 3x3 pad 1) */
    int32_t conv, conv_mode, Bn, H, W, Cin, Ho, Wo;
    /* optional fp32 workspace for split-K (small-M, long-K problems); NULL disables split-K */
    void* workspace; int64_t workspace_bytes;
    /* optional second output (NULL = none): per-row-chunk sums of the STORED fp16 values for the GroupNorm that consumes C
     * (norm1 / norm2 of ResnetBlock2D, Transformer2DModel.norm): gn_stats[(m / rows) * (N / 10) + n / 10][2] = (sum, sum of squares) over
     * the rows [rows * (m / rows), ...) of C and the 10 channels [10 * (n / 10), ...) -- 10 divides every group width of the SD-v1.5 U-Net
     * (10 / 20 / 30 / 40 / 60 / 80 channels), so one buffer serves a tensor both as a GroupNorm's only input and as one half of a
     * channel concatenation.  ``rows`` = fd_gemm_stats_rows() (32 for every kernel that can write the statistics, formed by one canonical
     * procedure per chunk: they do not depend on the tile policy or the batch size; 0 = this problem's kernel cannot): size the buffer
     * ceil(M / rows) * (N / 10) * 2 floats; fd_gemm fails if gn_stats is set and the chosen kernel cannot write it. */
    float* gn_stats;
    /* optional per-column factor (colscale_cols == 0: none): columns n < colscale_cols of alpha * (A.B^T + A2.B2^T) are multiplied by ``colscale``
     * in fp32 before bias / activation / rounding (colscale_cols must be a multiple of 4).  Attention.to_q inside the stacked q/k/v projection:
     * q is written pre-multiplied by softmax_scale * log2(e), which the attention kernels then take as is (negative ``scale`` argument). */
    float colscale; int32_t colscale_cols;
} fd_gemm_desc;
int fd_gemm(const fd_gemm_desc* d, void* stream);
/* rows per statistics chunk of ``gn_stats`` for this problem (32), or 0 when the kernel fd_gemm would launch has no statistics epilogue
 * (split-K, the 64-column wave tiles, fp32 / GEGLU outputs, N not a multiple of 80) */
int fd_gemm_stats_rows(const fd_gemm_desc* d);
/* tile variant fd_gemm would pick for this problem, as BM*1000+BN (128128 / 128064 / 64064) */
int fd_gemm_tile(const fd_gemm_desc* d);
/* name of the kernel fd_gemm launches for this problem as rocprofv3 --kernel-trace spells it (host buffer ``buf`` of ``n`` bytes);
 * returns the split-K factor (>1: a splitk_reduce_kernel follows the GEMM).  Measurement aid for bench.py's roofline. */
int fd_gemm_kernel_name(const fd_gemm_desc* d, char* buf, int n);

/* Direct convolution for tiny channel counts (conv_in 4->320, conv_out dgrad, VAE post_quant 1x1,
 * classifier stem).  x: [B,Cin,H,W] (nchw!=0) or [B,H,W,Cin]; w: fp32 [k*k*Cin, Cout]; y: fp16 [B,Ho,Wo,Cout]. */
int fd_conv_small_cin(const void* x, int x_is_f32, int nchw, const float* w, const float* bias, void* y,
                      int B, int H, int W, int Cin, int Cout, int ksize, int stride, int act, void* stream);
/* y[B,C,HW] (fp32 or fp16) <- clamp(scale * x[B,HW,ldx>=C] fp16, lo, hi)  (eps for the scheduler; images.clamp(-1,1) :1059,:1134) */
int fd_nhwc_to_nchw(const void* x, int64_t ldx, void* y, int y_is_f32, int B, int HW, int C, float scale, float lo, float hi, void* stream);
/* backward of the clamp: dpre[B,C,HW] fp32 = (lo <= pre <= hi) ? dimg : 0 */
int fd_clamp_bwd(const void* pre, int64_t ldx, const float* dimg, float* dpre, int B, int HW, int C, float lo, float hi, void* stream);

/* ---- GroupNorm (32 groups, channels-last, optional 2-source channel concat, optional fused SiLU).
 * Replaces torch.nn.GroupNorm(+SiLU) in ResnetBlock2D / Transformer2DModel / conv_norm_out.          */
/* y = act(GN(x)); mean_rstd [B,groups,2] receives the statistics for the backward; scratch: B*64*groups*2 floats */
int fd_groupnorm_fwd(const void* x1, int C1, const void* x2, int C2, int B, int HW, int groups, float eps, const float* gamma,
                     const float* beta, int silu, void* y, float* mean_rstd, float* scratch, void* stream);
/* the same with the statistics pass replaced by the producers' ``gn_stats`` buffers (fd_gemm_desc): st1 / st2 belong to x1 / x2 (st2 NULL
 * when C2 == 0), rows1 / rows2 are their chunk heights and must divide HW; C1, C2 and C / groups must be multiples of 10.  One launch,
 * x is read once. */
int fd_groupnorm_fwd_stats(const void* x1, int C1, const void* x2, int C2, int B, int HW, int groups, float eps, const float* gamma,
                           const float* beta, int silu, void* y, float* mean_rstd, const float* st1, int rows1, const float* st2, int rows2,
                           void* stream);
/* the same for inputs whose statistics arrive in the phase-major chunk order of FD_CONV_UP2PI: per1 / per2 = chunks per image AND phase of st1 / st2
 * (HW / rows / 4), or 0 for the plain image-major order */
int fd_groupnorm_fwd_stats_p(const void* x1, int C1, const void* x2, int C2, int B, int HW, int groups, float eps, const float* gamma,
                             const float* beta, int silu, void* y, float* mean_rstd, const float* st1, int rows1, int per1, const float* st2,
                             int rows2, int per2, void* stream);
/* backward: dx = d/dx [ act(GN(x)) ] . dy ; writes the two channel slices to dx1/dx2, optionally adding add1/add2 */
int fd_groupnorm_bwd(const void* x1, int C1, const void* x2, int C2, const void* dy, int B, int HW, int groups,
                     const float* mean_rstd, const float* gamma, const float* beta, int silu,
                     float* scratch, const void* add1, const void* add2, void* dx1, void* dx2, void* stream);

/* ---- LayerNorm over the last dim of [M,C] (BasicTransformerBlock.norm1/2/3, CLIP layer norms). */
int fd_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean_rstd /* [M,2] or NULL */,
                     int M, int C, float eps, void* stream);
int fd_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* mean_rstd, const void* add /* or NULL */,
                     void* dx, int M, int C, void* stream);

/* ---- elementwise */
int fd_geglu_fwd(const void* proj /* [M,2F] */, void* y /* [M,F] */, int M, int F, void* stream);
int fd_geglu_bwd(const void* proj, const void* dy, void* dproj, int M, int F, void* stream);
/* same with (value_c, gate_c) adjacent in proj / dproj [M, 2F] (the layout the fused FD_ACT_GEGLU projection keeps) */
int fd_geglu_bwd_interleaved(const void* proj, const void* dy, void* dproj, int M, int F, void* stream);
int fd_act_fwd(const void* x, void* y, int64_t n, int act, void* stream);
int fd_act_bwd(const void* z, const void* dy, void* dx, int64_t n, int act, void* stream);
int fd_add(const void* a, const void* b, void* y, int64_t n, float sa, float sb, void* stream);      /* y = sa*a + sb*b (fp16) */
int fd_copy_cols(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t M, int cols, void* stream); /* strided 2-D copy, fp16 */
int fd_transpose_btc(const void* x /* [B,T,C], row stride ldx (0 = C) */, int64_t ldx, void* y /* [B,C,Tp] */, int B, int T, int C, int Tp, void* stream);
int fd_downsum2x2(const void* x /* [B,2H,2W,C] */, void* y /* [B,H,W,C] */, int B, int H, int W, int C, void* stream);
/* y = softmax(scale*x + mask); mask fp32 [.., mask_t, cols], mask row = (row / mask_ht) * mask_t + row % mask_t, or NULL */
int fd_softmax_rows(const void* x, void* y, int64_t rows, int cols, float scale, const float* mask, int mask_t, int mask_ht, void* stream);
int fd_softmax_rows_bwd(const void* p, const void* dp, void* ds, int64_t rows, int cols, float scale, void* stream);
int fd_cast_f32_to_f16(const float* x, void* y, int64_t n, float scale, void* stream);
int fd_cast_f16_to_f32(const void* x, float* y, int64_t n, float scale, void* stream);

/* ---- fused attention (diffusers Attention.get_attention_scores + bmm, LoRAAttnProcessor.__call__)
 * q:[B,Tq,H*d]  k:[Bk,Tkr,H*d] of which the first Tk rows are keys (Tkr>=Tk: row-padded token buffers of the ViTs)
 * v: [Bk,Tkr,H*d] rows of stride ldk, exactly like k -- consumed as the projection wrote it, through LDS transpose reads (ds_read_b64_tr_b16); the
 *    transposed-copy operands (vt, kt, qt, d_ot) of ABI <= 3 are gone.
 * sample b uses kv batch b / kv_div (cross-attention K/V are shared by each CFG half).
 * o:[B,Tq,H*d] fp16, lse:[B,H,Tq] fp32 (natural-log sum-exp of the scaled scores).
 * scale: the softmax scale (d^-0.5).  NEGATIVE scale, all three entry points (d % 16 == 8 -- the U-Net's d = 40):
 *   "pre-scaled q" -- q holds q_true * |scale| * log2(e), written that way by its projection (fd_gemm_desc.colscale); the kernels then take the QK^T
 *   accumulator as the exponent's argument, the softmax reference point (forward) or the saved log-sum-exp (backward) riding in spare contraction
 *   slots of the padded head dim.  o, lse, dk, dv are unchanged in meaning and dq is still the gradient w.r.t. q_true.
 * ldq / ldk (and ldkv, lddq, lddkv below): row strides in elements of q, k/v and of the dq, dk/dv outputs; 0 = H*d (contiguous).
 * Non-trivial strides let q, k, v (and dq, dk, dv) be column slices of ONE [M, 3*H*d] buffer: the self-attention projections run as a
 * single GEMM with stacked weights and their input gradients as a single GEMM over K = 3*H*d.                                       */
int fd_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Tq, int Tk,
                int Tkr, int d, int kv_div, float scale, int ldq, int ldk, void* stream);
/* D[b,h,t] = sum_j dO*O */
int fd_attn_bwd_prep(const void* o, const void* d_o, float* D, int B, int H, int T, int d, void* stream);
/* dq from (q, k, v, dO, lse, D): K^T for the dS.K product comes from the row-major K tile through LDS transpose reads.
 * With o != NULL the kernel computes D = rowsum(dO*O) itself and WRITES it to D
 * for fd_attn_bwd_dkdv (fd_attn_bwd_prep is then not needed); with o == NULL it reads D. */
int fd_attn_bwd_dq(const void* q, const void* k, const void* v, const void* d_o, const float* lse,
                   float* D, const void* o, void* dq, int B, int H, int Tq, int Tk, int Tkr, int d, int kv_div, float scale,
                   int ldq, int ldkv, int lddq, void* stream);
/* dk,dv from (q, k, v, dO, lse, D): Q^T and dO^T come from the row-major q / dO tiles through LDS transpose reads.
 * When kv_div>1 the kv batch is shared by
 * kv_div consecutive samples; dk/dv are then fp32 [Bk,Tk,H*d] accumulated with atomics, else fp16 (overwritten).
 * ``accumulate`` == 1 selects the fp32-atomic form at kv_div == 1 too: several launches -- timesteps of the truncated chain whose
 * backwards run on different HIP streams -- may then add into one accumulator concurrently.
 * ``accumulate`` == 2 (round 4, what the training step uses): NO atomics -- dk / dv are fp32 [kv_div][Bk*Tkr][lddkv] and sample j of every
 * K/V group WRITES slab j; fd_sum_slabs then adds the slabs in a fixed order: shared dK / dV are bit-reproducible. */
int fd_attn_bwd_dkdv(const void* q, const void* k, const void* v, const void* d_o,
                     const float* lse, const float* D, void* dk, void* dv, int B, int H, int Tq, int Tk, int Tkr, int d,
                     int kv_div, float scale, int ldq, int ldkv, int lddkv, int accumulate, void* stream);
/* out[i] = sum_{s < nslab} in[s*n + i], s ascending (fp32): the fixed-order reduction behind accumulate == 2 */
int fd_sum_slabs(const float* in, float* out, int nslab, int64_t n, void* stream);

/* ---- the cross-attention sub-block of BasicTransformerBlock as ONE launch (diffusers attention.py BasicTransformerBlock.forward: norm2 -> attn2 ->
 * residual -> norm3; attn2 runs the attention processor selected at exp-1 main:811-817): the frozen rollout R2 (exp-1 main:1844-1858) without LoRA slabs or
 * recording, the finetuned model's rollout with both (descriptor fields below):
 *     n2 = LayerNorm(x; ln2);  q = n2 . wq^T;  o = softmax(q k^T * scale) v;  y = o . wo^T + bo + x;  yn = LayerNorm(y; ln3)
 * x, y, yn: [M, C] working dtype, contiguous; wq, wo: [C, C] (out, in); k: [Bk*L, C] (the L <= 80 prompt tokens per sample, as fd_attn_fwd takes
 * them); vt: [Bk, C, Lp] = V transposed with zero-padded keys (fd_transpose_btc, Lp >= 80); row m belongs to sample m / rows_per_sample, which reads
 * K / V batch (m / rows_per_sample) / kv_div.  C in {320, 640} with 8 heads (the 64^2 / 32^2 levels; wider levels keep the separate launches); M and rows_per_sample multiples of 64.
 * yn (and ln3_*) may be NULL; yn_stats [M, 2] = (mean, rstd) of LayerNorm3 or NULL.  Replaces fd_layernorm_fwd + fd_gemm + fd_attn_fwd + fd_gemm +
 * fd_layernorm_fwd; y, yn are bit-identical to that sequence's wherever its attention output o is (the softmax here rounds q once, pre-scaled). */
typedef struct fd_cross_block_desc {
    int32_t struct_size;             /* sizeof(fd_cross_block_desc) */
    const void* x;
    const float* ln2_gamma; const float* ln2_beta; float ln2_eps;
    const void* wq;
    const void* k; const void* vt; int32_t L, Lp;
    const void* wo; const float* bo;
    const float* ln3_gamma; const float* ln3_beta; float ln3_eps;
    void* y; void* yn; float* yn_stats;
    int32_t M, C, heads, rows_per_sample, kv_div;
    float scale;                     /* softmax scale, d^-0.5 */
    /* optional LoRA slabs of attn2.to_q / attn2.to_out (LoRAAttnProcessor.__call__: to_q(h) + scale * to_q_lora(h), exp-1 main:811-817), all four or none:
     * down [rp, C] and up [C, rp] in the working dtype, the LoRA scale folded into up (as fd_lora_refresh_multi writes them), rp = rank padded to 8 or 16 */
    const void* lora_q_down; int64_t ld_q_down; const void* lora_q_up; int64_t ld_q_up;
    const void* lora_o_down; int64_t ld_o_down; const void* lora_o_up; int64_t ld_o_up;
    int32_t lora_rp;
    /* optional recording for the backward (n2_out != NULL turns it on; then all of these except tq_out / to_out without LoRA are required):
     * n2_out [M, C] = LayerNorm2(x), ln2_stats [M, 2]; q_out [M, C] = the query as fd_attn_bwd_* take it (times scale * log2(e) when q_prescaled, see
     * fd_attn_fwd); tq_out [M, rp] = n2 . down_q^T; o_out [M, C] attention output; lse_out [M / rows_per_sample, heads, rows_per_sample]; to_out [M, rp] = o . down_o^T.
     * Rounding contract at head dim 80 (C = 640), where q cannot be stored pre-scaled: the kernel's own softmax runs on q rounded ONCE after the multiplication by
     * scale * log2(e), while q_out is the fp16 rounding of the UNSCALED accumulator -- two roundings of one fp32 value.  fd_attn_bwd_* recompute the probabilities
     * from q_out against lse_out, so their rows sum to 1 only to ~1e-3 relative per score (the separate launches round q once and use that in both directions).
     * Inside the tolerances of the parity tests (test_cross_attn_block_with_lora_slabs_and_recording compares dq / dK / dV of the fused forward at C = 640 with
     * fp32 torch); at head dim 40 (C = 320) forward and backward see the same stored q. */
    void* n2_out; float* ln2_stats; void* q_out; void* tq_out; void* o_out; float* lse_out; void* to_out;
    int32_t q_prescaled;
} fd_cross_block_desc;
int fd_cross_attn_block(const fd_cross_block_desc* d, void* stream);

/* ---- masked attention of the CLIP text encoder (transformers CLIPAttention; reference call sites :1011-1014, :1078-1081).
 * q,k,v,o: [B,T,H*d] fp16, T<=128, d<=128; key_valid [B,T] int32 or NULL; P [B,H,T,T] fp32 (saved probabilities) or NULL */
int fd_small_attn_fwd(const void* q, const void* k, const void* v, void* o, float* P, const int32_t* key_valid, int B, int H, int T,
                      int d, float scale, int causal, void* stream);
int fd_small_attn_bwd(const void* q, const void* k, const void* v, const float* P, const void* d_o, void* dq, void* dk, void* dv,
                      int B, int H, int T, int d, float scale, void* stream);

/* ---- LoRA weight gradients: G[n, r] (+)= sum_m X[m, n] * T[m, r]  (fp16 X,T; fp32 G with strides) */
int fd_lora_wgrad(const void* X, int64_t ldx, const void* T, int64_t ldt, float* G, int64_t g_stride_n, int64_t g_stride_r,
                  int M, int N, int R, float scale, float* scratch, int64_t scratch_elems, void* stream);

/* batched form: n <= FD_WGRAD_MAX independent problems (HOST array of descriptors) that share the padded rank (R <= 8, or 9..16) in one
 * partial + one final launch -- the 16 LoRA weight gradients of one transformer block's backward.  Same arithmetic and fixed reduction order
 * per problem as fd_lora_wgrad.  Requires N %% 8 == 0 and ldx %% 8 == 0 (R <= 8) or %% 4 (R <= 16). */
#define FD_WGRAD_MAX 16
typedef struct fd_wgrad_desc {
    int32_t struct_size;                 /* sizeof(fd_wgrad_desc), checked for every element (see fd_gemm_desc) */
    const void* X; int64_t ldx;          /* fp16 [M, N], row stride ldx */
    const void* T; int64_t ldt;          /* fp16 [M, RP] (rank padded), row stride ldt */
    float* G; int64_t g_stride_n, g_stride_r;   /* G[n * g_stride_n + r * g_stride_r] += scale * sum_m X[m,n] T[m,r] */
    int32_t M, N, R;
    float scale;
} fd_wgrad_desc;
int fd_lora_wgrad_multi(const fd_wgrad_desc* descs, int n, float* scratch, int64_t scratch_elems, void* stream);

/* Operand copies of LoRA pairs after an optimiser step (1-main-debias.py:2016-2029 leaves fp32 parameters; the MFMA slabs want 16-bit, rank-padded,
 * and transposed copies): for each pair  d16[j, k] = down[j, k] (rows j >= r zero),  dT16[k, j] = d16[j, k],  u16[n, j] = scale * up[n, j] (columns
 * j >= r zero),  uT16[j, n] = u16[n, j].  The four outputs are written through row strides, so that several pairs can live inside one stacked
 * buffer (the fused q/k/v projection of a self-attention layer).  Any number of pairs per call (chunked internally). */
typedef struct fd_lora_refresh_desc {
    int32_t struct_size;                           /* sizeof(fd_lora_refresh_desc), checked for every element (see fd_gemm_desc) */
    const float* down; const float* up;            /* fp32 [r, K], [N, r] */
    void* d16; int64_t ld_d16;                     /* 16-bit [rp, K] */
    void* dT16; int64_t ld_dT16;                   /* 16-bit [K, rp] */
    void* u16; int64_t ld_u16;                     /* 16-bit [N, rp] */
    void* uT16; int64_t ld_uT16;                   /* 16-bit [rp, N] */
    int32_t r, rp, K, N;
    float scale;
} fd_lora_refresh_desc;
int fd_lora_refresh_multi(const fd_lora_refresh_desc* descs, int n, void* stream);

/* ---- scheduler / CFG (DPMSolverMultistepScheduler.step + CFG combine, 1-main-debias.py:1051-1056,1123-1131)
 * eps:[2N,4,HW] fp32 NCHW (uncond first); x0_prev/x0_out fp32; lat fp32 updated in place.
 * x0 = (lat - sigma*e)/alpha ; lat' = c_x*lat - c_d0*x0 - c_d1*(x0 - x0_prev)                        */
int fd_cfg_dpm_step(const float* eps, float guidance, float* lat, const float* x0_prev, float* x0_out,
                    float alpha_t, float sigma_t, float c_x, float c_d0, float c_d1, int64_t n, void* stream);

/* ---- optimizer: flat fp32 buffers (1-main-debias.py:1998-2029) */
int fd_grad_finite_scale(float* g, int64_t n, float scale, int32_t* nonfinite_flag, void* stream);
int fd_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int32_t step, float ema_one_minus_decay, void* stream);

/* ---- classifier pieces (torchvision mobilenet_v3_large, 1-main-debias.py:929-935,1369) */
int fd_dwconv_fwd(const void* x, const float* w /* [k,k,C] */, const float* bias, void* y, int B, int H, int W, int C,
                  int k, int stride, int act, void* stream);
int fd_dwconv_bwd(const void* dy, const float* w, void* dx, int B, int H, int W, int C, int k, int stride, void* stream);
int fd_avgpool_hw(const void* x, void* y, int B, int HW, int C, void* stream);              /* [B,HW,C] -> [B,C] */
int fd_avgpool_hw_bwd(const void* dy /* [B,C] */, const void* add /* [B,HW,C] or NULL */, void* dx, int B, int HW, int C, void* stream);
int fd_scale_channels(const void* x, const void* s, void* y, int B, int HW, int C, void* stream); /* y = x * s[b,c] */
int fd_scale_channels_bwd(const void* x, const void* s, const void* dy, void* dx, void* ds /* fp16 [B,C] */, int B, int HW, int C, void* stream);
int fd_conv_small_cin_bwd(const void* dy /* [B,Ho,Wo,Cout] */, const float* w, float* dx /* fp32 [B,Cin,H,W], overwritten */,
                          int B, int H, int W, int Cin, int Cout, int ksize, int stride, float scale, void* stream);
/* crop [x0,y0,x1,y1) (may exceed the image -> fill) + bilinear resize (align_corners=False, no antialias):
 * img [B,3,H,W] fp16 NCHW -> chips [B,3,S,S] fp16 NCHW. */
int fd_crop_resize_fwd(const void* img, const int32_t* boxes, float fill, void* chips, int B, int H, int W, int S, void* stream);
int fd_crop_resize_bwd(const float* dchips /* [B,3,S,S] */, const int32_t* boxes, float* dimg /* [B,3,H,W] (+=) */,
                       int B, int H, int W, int S, void* stream);

/* ---- image-semantics regularisers (get_clip_feat / get_dino_feat, exp-1 1-main-debias.py:1139-1175; Resize :1860,1905).
 * chips [N,3,S,S] fp16 NCHW in [-1,1] -> patches [N*(S/P)^2, Kp] fp16 = ((x+1)/2 - mean_c)/std_c in Conv2d(3,D,P,P) weight order
 * (k = c*P*P + py*P + px), zero-padded to Kp; the patch embedding is then a GEMM.  bwd: dchips fp32 (+)= 0.5/std_c*scale*dpatches */
int fd_patchify_fwd(const void* chips, void* patches, const float* mean3, const float* std3, int N, int S, int P, int Kp, void* stream);
int fd_patchify_bwd(const void* dpatches, float* dchips, const float* std3, int N, int S, int P, int Kp, float scale, int accumulate,
                    void* stream);
/* face alignment of the face-realism term (image_pipeline :292-312: skimage SimilarityTransform + kornia.warp_affine, bilinear,
 * zeros padding in 0..255 space == -1 in [-1,1] space).  A [n,6] fp32: row-major 2x3 map from output pixel (x,y,1) to the input
 * sampling position in pixels; src_index [n]: image of chip k.  chips [n,3,S,S] fp16; bwd accumulates into dimg [B,3,H,W] fp32
 * as a fixed-order gather (one thread per image pixel, chips in ascending k: bit-reproducible, no atomics). */
int fd_warp_affine_fwd(const void* img, const int32_t* src_index, const float* A, float fill, void* chips, int n_chips, int H, int W,
                       int S, void* stream);
int fd_warp_affine_bwd(const float* dchips, const int32_t* src_index, const float* A, float* dimg, int n_chips, int B, int H, int W,
                       int S, void* stream);
/* apply_grad_hook_face (:1584-1617) in the backward: dimg [B,3,H,W] fp32 *= factors[b] inside rects[b] = [x0,y0,x1,y1) */
int fd_rect_scale(float* dimg, const int32_t* rects, const float* factors, int B, int H, int W, void* stream);

/* ---- dynamic targets of the multi-attribute experiments: the Monte-Carlo transport solves (exp-3-debias-gender-race/1-main-debias.py:1488-1536
 * ``ot.emd(ones(N), counts_s, M)`` for s < S; exp-4 :1517-1569).  N unit-mass faces onto K cells whose integer capacities counts[s, :]
 * sum to N: an assignment problem on the capacity-replicated cost matrix, solved exactly (shortest augmenting paths, fp64) by one
 * wave per draw.  cost [N,K] f64, counts [S,K] int32, plan [N,K] f32: plan += sum_s T_s (0/1 transport matrices; zero it first for a
 * fresh sum); seats [S,N] int32 (cell of face i in draw s) or NULL.  N <= 1024. */
int fd_ot_assign_sum(const double* cost, const int32_t* counts, float* plan, int32_t* seats, int N, int K, int S, void* stream);

#ifdef __cplusplus
}
#endif
#endif
