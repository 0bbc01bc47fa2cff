// FP8 (OCP e4m3fn) self-attention forward for gfx950 -- BASELINE configs[4] ("bf16 + MFMA fp8 attention"; SURVEY 8d: fp8 e4m3,
// per-tile scaled QK^T / PV in self-attention only).  The reference never ran this (it only ever used fp16,
// exp-1-debias-gender/1-main-debias.py:401-405); it replaces the same diffusers call as fd_attn_fwd
// (Attention.get_attention_scores + bmm inside LoRAAttnProcessor.__call__, injected at :798-818) for the attn1 layers.
//
// Two kernels:
//   attn_fp8_quant_kv   K [B,T,C] / V [B,T,C] (working dtype) -> K8 [B,H,T,DK8] and V8^T [B,H,DV8,T] in e4m3 with one scale per
//                       (b, h, 64-key tile) each (amax / 448), zero padded to the MFMA shapes (DK8 = d up to x16, DV8 = d up to x32)
//   attn_fwd_fp8        the flash-style forward of attn.hip with both contractions on v_mfma_f32_32x32x16_fp8_fp8: Q is quantised in
//                       registers with one scale per query row; the K-tile scale multiplies the raw scores; P (<= 1) is quantised as
//                       p * s_v[tile] * 256 / max_t s_v so that the V-tile scale rides in P and one accumulator serves all tiles.
// Softmax statistics, the output normalisation and LSE are fp32; O is written in the working dtype and LSE in the same convention as
// fd_attn_fwd, so the (bf16) backward kernels of attn.hip consume them unchanged.
#include "common.h"

#define LOG2E 1.4426950408889634f
#define FP8_MAX 448.f

typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16_t zero16f() {
    f32x16_t z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
__device__ __forceinline__ int crow8(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

// two floats -> two e4m3 bytes in the low (hi = false) or high half of a 32-bit word
__device__ __forceinline__ uint32_t pk_fp8(float a, float b, uint32_t old, bool hi) {
    return hi ? (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, true) : (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, false);
}
__device__ __forceinline__ uint32_t pk4_fp8(float a, float b, float c, float d) { return pk_fp8(c, d, pk_fp8(a, b, 0u, false), true); }

// ================================================================================== quantisation pre-pass
// grid (T/64, H, B), 256 threads.  K tile: thread t owns key t>>2, d-quarter t&3.  V tile is transposed through LDS.
template <int D>
__global__ __launch_bounds__(256) void attn_fp8_quant_kv_kernel(const f16* __restrict__ K, const f16* __restrict__ V, uint8_t* __restrict__ K8,
                                                                uint8_t* __restrict__ V8t, float* __restrict__ SK, float* __restrict__ SV, int H,
                                                                int T, int ldkv) {
    constexpr int DK8 = (D + 15) / 16 * 16, DV8 = (D + 31) / 32 * 32;
    __shared__ float red[8];
    __shared__ float vt[64][D + 1];
    const int tile = blockIdx.x, h = blockIdx.y, b = blockIdx.z, nT = gridDim.x;
    const int tid = threadIdx.x, key = tid >> 2, part = tid & 3;
    const int t = tile * 64 + key;
    // ---- K: per-thread slice of one key row
    constexpr int PER = (DK8 / 4);                          // d columns per thread (12 / 20 / 40)
    float kv[PER];
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = part * PER + j;
        kv[j] = (t < T && c < D) ? (float)K[((int64_t)b * T + t) * ldkv + h * D + c] : 0.f;
        amax = fmaxf(amax, fabsf(kv[j]));
    }
    amax = wave_max(amax);
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sk = amax > 0.f ? amax / FP8_MAX : 1.f, isk = 1.f / sk;
    uint8_t* kdst = K8 + (((int64_t)b * H + h) * T + t) * DK8 + part * PER;
    if (t < T) {
#pragma unroll
        for (int j = 0; j < PER; j += 4) *(uint32_t*)(kdst + j) = pk4_fp8(kv[j] * isk, kv[j + 1] * isk, kv[j + 2] * isk, kv[j + 3] * isk);
    }
    // ---- V: load rows, amax, transpose via LDS, write [DV8][64] bytes of this tile
    float vmax = 0.f;
    for (int i = tid; i < 64 * D; i += 256) {
        const int r = i / D, c = i - r * D;
        const float x = (tile * 64 + r < T) ? (float)V[((int64_t)b * T + tile * 64 + r) * ldkv + h * D + c] : 0.f;
        vt[r][c] = x;
        vmax = fmaxf(vmax, fabsf(x));
    }
    vmax = wave_max(vmax);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = vmax;
    __syncthreads();
    vmax = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    const float sv = vmax > 0.f ? vmax / FP8_MAX : 1.f, isv = 1.f / sv;
    if (tid == 0) {
        SK[((int64_t)b * H + h) * nT + tile] = sk;
        SV[((int64_t)b * H + h) * nT + tile] = sv;
    }
    // thread -> (dv row, 16 keys): DV8 rows x 4 key-quarters
    for (int i = tid; i < DV8 * 4; i += 256) {
        const int dv = i >> 2, kq = (i & 3) * 16;
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = dv < D ? vt[kq + j * 4 + e][dv] * isv : 0.f;
            w[j] = pk4_fp8(x[0], x[1], x[2], x[3]);
        }
        uint8_t* dst = V8t + (((int64_t)b * H + h) * DV8 + dv) * T + tile * 64 + kq;
        *(uint4*)dst = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ================================================================================== forward
// grid (Tq/128, H, B), 4 waves x 32 queries; lane = (q = lane & 31, g = lane >> 5) exactly as attn_fwd_kernel.
template <int D>
__global__ __launch_bounds__(256) void attn_fwd_fp8_kernel(const f16* __restrict__ Q, const uint8_t* __restrict__ K8, const uint8_t* __restrict__ V8t,
                                                           const float* __restrict__ SK, const float* __restrict__ SV, f16* __restrict__ O,
                                                           float* __restrict__ LSE, int H, int T, float scale, int ldq) {
    constexpr int DK8 = (D + 15) / 16 * 16, DV8 = (D + 31) / 32 * 32;
    constexpr int NKS = DK8 / 16, NDV = DV8 / 32;
    constexpr int KLD = DK8 + 8;          // LDS row strides in bytes: conflict-free for the 8-byte / 4-byte fragment reads
    constexpr int VLD = 68;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem8[];
    uint8_t* Ks = smem8;                  // [64][KLD]
    uint8_t* Vts = smem8 + 64 * KLD;      // [DV8][VLD]
    __shared__ float svmax_s;

    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 128;
    const int C = H * D, nT = T / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 31, g = lane >> 5;
    const int t = q0 + wave * 32 + ql;
    const bool tvalid = t < T;

    // ---- Q fragment: 8 values per k-step for this lane's query row, quantised with one scale per row
    float qv[NKS][8];
    float qmax = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int col = ks * 16 + g * 8;
        f16x8 x = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (tvalid && col < D) x = *(const f16x8*)(Q + ((int64_t)b * T + t) * ldq + h * D + col);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            qv[ks][j] = (float)x[j];
            qmax = fmaxf(qmax, fabsf(qv[ks][j]));
        }
    }
    qmax = fmaxf(qmax, __shfl_xor(qmax, 32, 64));
    const float sq = qmax > 0.f ? qmax / FP8_MAX : 1.f, isq = 1.f / sq;
    long qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const uint32_t lo = pk4_fp8(qv[ks][0] * isq, qv[ks][1] * isq, qv[ks][2] * isq, qv[ks][3] * isq);
        const uint32_t hi = pk4_fp8(qv[ks][4] * isq, qv[ks][5] * isq, qv[ks][6] * isq, qv[ks][7] * isq);
        qf[ks] = (long)(((uint64_t)hi << 32) | lo);
    }
    // ---- V scale reference: max over the tiles of this (b, h)
    const float* skp = SK + ((int64_t)b * H + h) * nT;
    const float* svp = SV + ((int64_t)b * H + h) * nT;
    if (wave == 0) {
        float m = 0.f;
        for (int i = lane; i < nT; i += 64) m = fmaxf(m, svp[i]);
        m = wave_max(m);
        if (lane == 0) svmax_s = m;
    }
    __syncthreads();
    const float svmax = svmax_s;
    const float cP = 256.f / svmax;          // P is stored as p * s_v[tile] * cP  (<= 256)

    f32x16_t oacc[NDV];
#pragma unroll
    for (int i = 0; i < NDV; ++i) oacc[i] = zero16f();
    float m_run = -INFINITY, l_run = 0.f;
    const float sl2q = scale * LOG2E * sq;   // raw fp8 score -> log2-domain logit, without the K-tile scale

    const uint8_t* Kb = K8 + ((int64_t)b * H + h) * (int64_t)T * DK8;
    const uint8_t* Vb = V8t + ((int64_t)b * H + h) * (int64_t)DV8 * T;

    // staging: K tile = 64*DK8 contiguous bytes (8-byte pieces: 64*DK8/8 of them), V tile = DV8 rows x 64 bytes (16-byte pieces, 4 per row)
    constexpr int KP = 64 * DK8 / 8, KPT = (KP + 255) / 256;     // 8-byte pieces per thread
    constexpr int VP = DV8 * 4, VPT = (VP + 255) / 256;          // 16-byte pieces per thread
    uint64_t kreg[KPT];
    uint4 vreg[VPT];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int p = threadIdx.x + i * 256;
            kreg[i] = p < KP ? *(const uint64_t*)(Kb + (int64_t)k0 * DK8 + (int64_t)p * 8) : 0ull;
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int p = threadIdx.x + i * 256;
            vreg[i] = p < VP ? *(const uint4*)(Vb + (int64_t)(p >> 2) * T + k0 + (p & 3) * 16) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int p = threadIdx.x + i * 256;
            if (p < KP) {
                const int r = (p * 8) / DK8, c = (p * 8) - r * DK8;
                *(uint64_t*)(Ks + r * KLD + c) = kreg[i];
            }
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int p = threadIdx.x + i * 256;
            if (p < VP) {
                uint8_t* d = Vts + (p >> 2) * VLD + (p & 3) * 16;
                *(uint32_t*)(d) = vreg[i].x; *(uint32_t*)(d + 4) = vreg[i].y; *(uint32_t*)(d + 8) = vreg[i].z; *(uint32_t*)(d + 12) = vreg[i].w;
            }
        }
    };

    load_tile(0);
    for (int k0 = 0; k0 < T; k0 += 64) {
        __syncthreads();
        store_tile();
        __syncthreads();
        f32x16_t s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            s[kt] = zero16f();
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const long kf = *(const long*)(Ks + (kt * 32 + ql) * KLD + ks * 16 + g * 8);
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(kf, qf[ks], s[kt], 0, 0, 0);
            }
        }
        if (k0 + 64 < T) load_tile(k0 + 64);     // next tile's loads fly under the softmax and the PV MFMAs
        const int tile = k0 >> 6;
        const float c2 = sl2q * skp[tile];       // log2-domain logit = s_raw * c2
        const float pv = svp[tile] * cP;
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c2;             // c2 > 0: max commutes with the scaling
        const float m_new = fmaxf(m_run, mx);                    // running max in the log2 domain
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float rs = 0.f;
        long pf[4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c2, -m_new));
                rs += p[r];
                p[r] *= pv;
            }
            // element j of a 16-key step <-> key 4g + (j & 3) + 8 (j >> 2): accumulator registers r = 8*half + j of this sub-tile
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const uint32_t lo = pk4_fp8(p[hf * 8 + 0], p[hf * 8 + 1], p[hf * 8 + 2], p[hf * 8 + 3]);
                const uint32_t hi = pk4_fp8(p[hf * 8 + 4], p[hf * 8 + 5], p[hf * 8 + 6], p[hf * 8 + 7]);
                pf[kt * 2 + hf] = (long)(((uint64_t)hi << 32) | lo);
            }
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
                const uint8_t* row = Vts + (i * 32 + ql) * VLD + st * 16 + 4 * g;
                const uint32_t lo = *(const uint32_t*)(row), hi = *(const uint32_t*)(row + 8);
                const long vf = (long)(((uint64_t)hi << 32) | lo);
                oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vf, pf[st], oacc[i], 0, 0, 0);
            }
    }
    if (tvalid) {
        const float inv = 1.f / (l_run * cP);
        f16* Op = O + ((int64_t)b * T + t) * C + h * D;
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
        // natural-log sum-exp of the scaled scores (m_run is in the log2 domain)
        if (LSE && g == 0) LSE[((int64_t)b * H + h) * T + t] = (m_run + log2f(l_run)) / LOG2E;
    }
}

template <int D> static constexpr size_t fp8_lds() {
    return (size_t)64 * ((D + 15) / 16 * 16 + 8) + (size_t)((D + 31) / 32 * 32) * 68;
}

extern "C" int fd_attn_fp8_quant_kv(const void* k, const void* v, void* k8, void* v8t, float* sk, float* sv, int B, int H, int T, int d, int ldkv,
                                    void* stream) {
    FD_REQUIRE(B > 0 && H > 0 && T > 0 && (T & 63) == 0, "fd_attn_fp8_quant_kv: T must be a positive multiple of 64");
    if (ldkv <= 0) ldkv = H * d;
    dim3 grid(T / 64, H, B);
    switch (d) {
        case 40: hipLaunchKernelGGL(attn_fp8_quant_kv_kernel<40>, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)k, (const f16*)v, (uint8_t*)k8, (uint8_t*)v8t, sk, sv, H, T, ldkv); break;
        case 80: hipLaunchKernelGGL(attn_fp8_quant_kv_kernel<80>, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)k, (const f16*)v, (uint8_t*)k8, (uint8_t*)v8t, sk, sv, H, T, ldkv); break;
        case 160: hipLaunchKernelGGL(attn_fp8_quant_kv_kernel<160>, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)k, (const f16*)v, (uint8_t*)k8, (uint8_t*)v8t, sk, sv, H, T, ldkv); break;
        default: fd_set_error("fd_attn_fp8_quant_kv: head dim %d not built (40 / 80 / 160)", d); return FD_ERR_ARG;
    }
    return fd_check_launch("fd_attn_fp8_quant_kv");
}

extern "C" int fd_attn_fwd_fp8(const void* q, const void* k8, const void* v8t, const float* sk, const float* sv, void* o, float* lse, int B, int H,
                               int T, int d, float scale, int ldq, void* stream) {
    FD_REQUIRE(B > 0 && H > 0 && T > 0 && (T & 63) == 0, "fd_attn_fwd_fp8: T must be a positive multiple of 64");
    if (ldq <= 0) ldq = H * d;
    FD_REQUIRE((ldq & 7) == 0, "fd_attn_fwd_fp8: ldq %% 8");
    dim3 grid((T + 127) / 128, H, B);
#define CALL8(DD)                                                                                                                              \
    hipLaunchKernelGGL(attn_fwd_fp8_kernel<DD>, grid, dim3(256), fp8_lds<DD>(), (hipStream_t)stream, (const f16*)q, (const uint8_t*)k8,       \
                       (const uint8_t*)v8t, sk, sv, (f16*)o, lse, H, T, scale, ldq)
    switch (d) {
        case 40: CALL8(40); break;
        case 80: CALL8(80); break;
        case 160: CALL8(160); break;
        default: fd_set_error("fd_attn_fwd_fp8: head dim %d not built (40 / 80 / 160)", d); return FD_ERR_ARG;
    }
#undef CALL8
    return fd_check_launch("fd_attn_fwd_fp8");
}
