// Masked multi-head attention for the CLIP text encoder (sequence <= 128 tokens, head dim <= 128):
// causal mask + key padding mask, forward and backward.  The text encoder sees 2 sequences of
// ~13 tokens per rollout, so this is a latency kernel: one workgroup per (batch, head), everything
// in LDS, fp32 math, no MFMA (the tiles would be >90 % padding).
#include "common.h"

#define SA_MAXT 128
#define SA_MAXD 128

// q,k,v,o: [B, T, H*d] fp16; key_valid: [B, T] int32 (1 = attend) or NULL; P (optional out): [B,H,T,T] fp32
__global__ __launch_bounds__(256) void small_attn_fwd_kernel(const f16* q, const f16* k, const f16* v, f16* o, float* Pout,
                                                             const int32_t* key_valid, int H, int T, int d, float scale, int causal) {
    extern __shared__ float sm[];
    float* ks = sm;                 // [T][d+1]
    float* vs = ks + T * (d + 1);   // [T][d+1]
    float* ps = vs + T * (d + 1);   // [4 waves][SA_MAXT]
    float* qs = ps + 4 * SA_MAXT;   // [4 waves][SA_MAXD]
    const int b = blockIdx.y, h = blockIdx.x, C = H * d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < T * d; i += 256) {
        const int t = i / d, c = i - t * d;
        ks[t * (d + 1) + c] = (float)k[((int64_t)b * T + t) * C + h * d + c];
        vs[t * (d + 1) + c] = (float)v[((int64_t)b * T + t) * C + h * d + c];
    }
    __syncthreads();
    for (int i0 = 0; i0 < T; i0 += 4) {   // uniform trip count: block barriers order the per-wave LDS scratch
        const int i = i0 + wave;
        const bool act = i < T;
        if (act)
            for (int c = lane; c < d; c += 64) qs[wave * SA_MAXD + c] = (float)q[((int64_t)b * T + i) * C + h * d + c];
        __syncthreads();
        float s[2];
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lane + u * 64;
            s[u] = -INFINITY;
            if (act && j < T) {
                bool ok = (!causal || j <= i) && (!key_valid || key_valid[b * T + j] != 0);
                if (ok) {
                    float a = 0.f;
                    for (int c = 0; c < d; ++c) a += qs[wave * SA_MAXD + c] * ks[j * (d + 1) + c];
                    s[u] = a * scale;
                }
            }
            mx = fmaxf(mx, s[u]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            s[u] = (s[u] == -INFINITY) ? 0.f : __expf(s[u] - mx);
            sum += s[u];
        }
        sum = wave_sum(sum);
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lane + u * 64;
            if (act && j < T) {
                ps[wave * SA_MAXT + j] = s[u] * inv;
                if (Pout) Pout[(((int64_t)b * H + h) * T + i) * T + j] = s[u] * inv;
            }
        }
        __syncthreads();
        if (act)
            for (int c = lane; c < d; c += 64) {
                float a = 0.f;
                for (int j = 0; j < T; ++j) a += ps[wave * SA_MAXT + j] * vs[j * (d + 1) + c];
                o[((int64_t)b * T + i) * C + h * d + c] = (f16)a;
            }
        __syncthreads();
    }
}

// backward from the saved probabilities P [B,H,T,T] fp32: dq, dk, dv [B,T,H*d] fp16
__global__ __launch_bounds__(256) void small_attn_bwd_kernel(const f16* q, const f16* k, const f16* v, const float* P, const f16* d_o,
                                                             f16* dq, f16* dk, f16* dv, int H, int T, int d, float scale) {
    extern __shared__ float sm[];
    float* qs = sm;                  // [T][d+1]
    float* ks = qs + T * (d + 1);
    float* vs = ks + T * (d + 1);
    float* gs = vs + T * (d + 1);    // dO
    float* dS = gs + T * (d + 1);    // [T][T+1]
    const int b = blockIdx.y, h = blockIdx.x, C = H * d;
    for (int i = threadIdx.x; i < T * d; i += 256) {
        const int t = i / d, c = i - t * d;
        const int64_t g = ((int64_t)b * T + t) * C + h * d + c;
        qs[t * (d + 1) + c] = (float)q[g];
        ks[t * (d + 1) + c] = (float)k[g];
        vs[t * (d + 1) + c] = (float)v[g];
        gs[t * (d + 1) + c] = (float)d_o[g];
    }
    __syncthreads();
    const float* Pb = P + ((int64_t)b * H + h) * T * T;
    // dP[i][j] = dO_i . v_j ; dS = P * (dP - sum_j P dP)   (one wave per row)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < T; i += 4) {
        float dp[2], pv[2];
        float dot = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lane + u * 64;
            dp[u] = pv[u] = 0.f;
            if (j < T) {
                pv[u] = Pb[i * T + j];
                float a = 0.f;
                for (int c = 0; c < d; ++c) a += gs[i * (d + 1) + c] * vs[j * (d + 1) + c];
                dp[u] = a;
                dot += pv[u] * a;
            }
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lane + u * 64;
            if (j < T) dS[i * (T + 1) + j] = pv[u] * (dp[u] - dot) * scale;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T * d; i += 256) {
        const int t = i / d, c = i - t * d;
        float aq = 0.f, ak = 0.f, av = 0.f;
        for (int j = 0; j < T; ++j) {
            aq += dS[t * (T + 1) + j] * ks[j * (d + 1) + c];
            ak += dS[j * (T + 1) + t] * qs[j * (d + 1) + c];
            av += Pb[j * T + t] * gs[j * (d + 1) + c];
        }
        const int64_t g = ((int64_t)b * T + t) * C + h * d + c;
        dq[g] = (f16)aq;
        dk[g] = (f16)ak;
        dv[g] = (f16)av;
    }
}

extern "C" int fd_small_attn_fwd(const void* q, const void* k, const void* v, void* o, float* P, const int32_t* key_valid, int B, int H, int T,
                                 int d, float scale, int causal, void* stream) {
    FD_REQUIRE(T >= 1 && T <= SA_MAXT && d >= 1 && d <= SA_MAXD, "fd_small_attn_fwd: T<=128, d<=128 (got T=%d d=%d)", T, d);
    const size_t lds = (size_t)(2 * T * (d + 1) + 4 * SA_MAXT + 4 * SA_MAXD) * sizeof(float);
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)small_attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    hipLaunchKernelGGL(small_attn_fwd_kernel, dim3(H, B), dim3(256), lds, (hipStream_t)stream, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)o,
                       P, key_valid, H, T, d, scale, causal);
    return fd_check_launch("fd_small_attn_fwd");
}

extern "C" int fd_small_attn_bwd(const void* q, const void* k, const void* v, const float* P, const void* d_o, void* dq, void* dk, void* dv, int B,
                                 int H, int T, int d, float scale, void* stream) {
    FD_REQUIRE(T >= 1 && T <= SA_MAXT && d >= 1 && d <= SA_MAXD, "fd_small_attn_bwd: T<=128, d<=128");
    const size_t lds = (size_t)(4 * T * (d + 1) + T * (T + 1)) * sizeof(float);
    FD_REQUIRE(lds <= 160 * 1024, "fd_small_attn_bwd: T=%d d=%d needs %zu B of LDS", T, d, lds);
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)small_attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    hipLaunchKernelGGL(small_attn_bwd_kernel, dim3(H, B), dim3(256), lds, (hipStream_t)stream, (const f16*)q, (const f16*)k, (const f16*)v, P,
                       (const f16*)d_o, (f16*)dq, (f16*)dk, (f16*)dv, H, T, d, scale);
    return fd_check_launch("fd_small_attn_bwd");
}
