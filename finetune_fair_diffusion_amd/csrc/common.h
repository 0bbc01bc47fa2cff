// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libfairdiff_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fairdiff_hip.h"

// The working dtype "wd" of the path: fp16 in the reference's configs 1-4 (`mixed_precision fp16`), bf16 in BASELINE
// configs[4].  One source, two libraries with the same C-ABI: libfairdiff_hip.so (fp16) and libfairdiff_hip_bf16.so (-DFD_BF16).
// ``f16`` is the name of that 16-bit storage type throughout the kernels; arithmetic is always fp32.
#ifdef FD_BF16
typedef __bf16 f16;
#define FD_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define FD_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define FD_WD_NAME "bf16"
#define FD_DOT2(a, b, c) __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false)     // v_dot2c_f32_bf16: c + a.x * b.x + a.y * b.y, fp32 accumulate
#else
typedef _Float16 f16;
#define FD_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define FD_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define FD_WD_NAME "fp16"
#define FD_DOT2(a, b, c) __builtin_amdgcn_fdot2(a, b, c, false)              // v_dot2c_f32_f16
#endif
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FD_WAVE 64

void fd_set_error(const char* fmt, ...);
int fd_check_launch(const char* what);

#define FD_REQUIRE(cond, ...)                  \
    do {                                       \
        if (!(cond)) {                         \
            fd_set_error(__VA_ARGS__);         \
            return FD_ERR_ARG;                 \
        }                                      \
    } while (0)

// every descriptor struct of the C-ABI starts with the caller's sizeof (include/fairdiff_hip.h): a binding compiled against another revision of the
// header is refused before any field is read
#define FD_REQUIRE_DESC(ptr, type, who)                                                                                                    \
    FD_REQUIRE((ptr) && (ptr)->struct_size == (int32_t)sizeof(type),                                                                        \
               who ": descriptor struct_size is %d but this library's " #type " has %d bytes (ABI revision %d): rebuild the binding against " \
                   "include/fairdiff_hip.h and set struct_size = sizeof(" #type ")",                                                         \
               (ptr) ? (int)(ptr)->struct_size : -1, (int)sizeof(type), FD_ABI_VERSION)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` holds >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float silu_grad_f(float x) {
    const float s = 1.f / (1.f + __expf(-x));
    return s * (1.f + x * (1.f - s));
}
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute): one reciprocal, one exponential and five FMAs, no branches -- the
// library erff costs several times that and sits in the GEGLU epilogue of every FF1 GEMM and in the GEGLU backward kernel
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = 1.f / (1.f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    return copysignf(1.f - poly * __expf(-ax * ax), x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.f + erf_as(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad_f(float x) {
    return 0.5f * (1.f + erf_as(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float quick_gelu_grad_f(float x) {
    const float s = 1.f / (1.f + __expf(-1.702f * x));
    return s * (1.f + 1.702f * x * (1.f - s));
}
__device__ __forceinline__ float hardswish_f(float x) { return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f); }
__device__ __forceinline__ float hardswish_grad_f(float x) {
    // sub-gradient convention at the kinks follows torch (2.10): 0 for x <= -3, x/3 + 0.5 inside, 1 for x >= 3
    if (x <= -3.f) return 0.f;
    if (x < 3.f) return x * (1.f / 3.f) + 0.5f;
    return 1.f;
}
__device__ __forceinline__ float hardsigmoid_f(float x) { return fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f); }

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case FD_ACT_SILU: return silu_f(v);
        case FD_ACT_QUICK_GELU: return quick_gelu_f(v);
        case FD_ACT_GELU: return gelu_erf_f(v);
        case FD_ACT_RELU: return fmaxf(v, 0.f);
        case FD_ACT_HARDSWISH: return hardswish_f(v);
        case FD_ACT_HARDSIGMOID: return hardsigmoid_f(v);
        default: return v;
    }
}
// derivative of act w.r.t. its pre-activation input z
__device__ __forceinline__ float act_grad(float z, int act) {
    switch (act) {
        case FD_ACT_SILU: return silu_grad_f(z);
        case FD_ACT_QUICK_GELU: return quick_gelu_grad_f(z);
        case FD_ACT_GELU: return gelu_erf_grad_f(z);
        case FD_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case FD_ACT_HARDSWISH: return hardswish_grad_f(z);
        case FD_ACT_HARDSIGMOID: return (z > -3.f && z < 3.f) ? (1.f / 6.f) : 0.f;
        default: return 1.f;
    }
}

// ---- workgroup trace (measurement builds only, -DFD_BENCH_HOOKS): with a log buffer installed (fd_bench_wg_trace, runtime.hip) every workgroup of an
// instrumented kernel appends (kernel id | waves << 8 | blockIdx.x << 16, XCC id << 32 | HW_ID, start, end) -- times on the 100 MHz s_memrealtime clock, which is
// common to the whole chip -- so that scratch/wg_fill.py can say how many CUs held work at any moment of the SHIPPED multi-stream schedule (rocprofv3's kernel
// trace serialises most dispatches: profiles/r05_kernel_trace_concurrency_start_of_round.txt).  One returning atomic + one 32-byte store per workgroup.
#ifdef FD_BENCH_HOOKS
static __device__ unsigned long long* fd_wgt_buf;      // [0] = record counter, records of 4 x u64 from [4] on; one copy per translation unit (FD_WGT_SETTER)
static __device__ unsigned int fd_wgt_cap;
struct FdWgTrace {
    unsigned long long t0;
    int id;
    __device__ __forceinline__ FdWgTrace(int id_) : t0(0), id(id_) {
        if (threadIdx.x == 0 && fd_wgt_buf) t0 = __builtin_amdgcn_s_memrealtime();
    }
    __device__ __forceinline__ ~FdWgTrace() {
        if (threadIdx.x == 0 && fd_wgt_buf) {
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            const unsigned long long i = atomicAdd(fd_wgt_buf, 1ULL);
            if (i < fd_wgt_cap) {
                const unsigned int hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
                unsigned long long* r = fd_wgt_buf + 4 + 4 * i;
                r[0] = (unsigned long long)id | ((unsigned long long)((blockDim.x * blockDim.y + 63) >> 6) << 8) | ((unsigned long long)blockIdx.x << 16);
                r[1] = ((unsigned long long)xcc << 32) | hw;
                r[2] = t0;
                r[3] = t1;
            }
        }
    }
};
#define FD_WG_TRACE(id) FdWgTrace fd_wg_trace_(id)
#define FD_WGT_SETTER(tu)                                                                  \
    extern "C" void fd_wgt_set_##tu(unsigned long long* buf, unsigned int cap) {            \
        (void)hipMemcpyToSymbol(HIP_SYMBOL(fd_wgt_buf), &buf, sizeof(buf));                 \
        (void)hipMemcpyToSymbol(HIP_SYMBOL(fd_wgt_cap), &cap, sizeof(cap));                 \
    }
#else
#define FD_WG_TRACE(id)
#define FD_WGT_SETTER(tu)
#endif

// XCD-aware bijective remap of a linear workgroup id: consecutive remapped ids share an XCD's L2
// (dispatcher places block b on XCD b % 8 -- speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
