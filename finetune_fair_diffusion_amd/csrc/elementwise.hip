// HBM-bound elementwise / layout kernels (16-byte fp16 vectors, grid-stride).
#include "common.h"

static inline dim3 grid_for(int64_t nvec, int threads = 256) {
    int64_t b = (nvec + threads - 1) / threads;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}

// ---------------------------------------------------------------- GEGLU
__global__ void geglu_fwd_kernel(const f16* proj, f16* y, int64_t M, int F) {
    const int FV = F >> 3;
    const int64_t n = M * FV;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / FV;
        const int f = (int)(i - m * FV) * 8;
        const f16x8 a = *(const f16x8*)(proj + m * 2 * F + f);
        const f16x8 g = *(const f16x8*)(proj + m * 2 * F + F + f);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)((float)a[j] * gelu_erf_f((float)g[j]));
        *(f16x8*)(y + m * F + f) = o;
    }
}
__global__ void geglu_bwd_kernel(const f16* proj, const f16* dy, f16* dproj, int64_t M, int F) {
    const int FV = F >> 3;
    const int64_t n = M * FV;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / FV;
        const int f = (int)(i - m * FV) * 8;
        const f16x8 a = *(const f16x8*)(proj + m * 2 * F + f);
        const f16x8 g = *(const f16x8*)(proj + m * 2 * F + F + f);
        const f16x8 d = *(const f16x8*)(dy + m * F + f);
        f16x8 da, dg;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gj = (float)g[j], dj = (float)d[j];
            da[j] = (f16)(dj * gelu_erf_f(gj));
            dg[j] = (f16)(dj * (float)a[j] * gelu_erf_grad_f(gj));
        }
        *(f16x8*)(dproj + m * 2 * F + f) = da;
        *(f16x8*)(dproj + m * 2 * F + F + f) = dg;
    }
}
__global__ void geglu_bwd_il_kernel(const f16* proj, const f16* dy, f16* dproj, int64_t n4) {
    FD_WG_TRACE(17);
    // 4 (value, gate) pairs per thread: 16 B of proj, 8 B of dy in, 16 B of dproj out
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f16x8 pr = *(const f16x8*)(proj + i * 8);
        const f16x4 d = *(const f16x4*)(dy + i * 4);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = (float)pr[2 * j], g = (float)pr[2 * j + 1], dj = (float)d[j];
            o[2 * j] = (f16)(dj * gelu_erf_f(g));
            o[2 * j + 1] = (f16)(dj * a * gelu_erf_grad_f(g));
        }
        *(f16x8*)(dproj + i * 8) = o;
    }
}
extern "C" int fd_geglu_bwd_interleaved(const void* proj, const void* dy, void* dproj, int M, int F, void* stream) {
    FD_REQUIRE((F & 3) == 0, "fd_geglu_bwd_interleaved: F %% 4");
    const int64_t n4 = (int64_t)M * (F / 4);
    hipLaunchKernelGGL(geglu_bwd_il_kernel, grid_for(n4), dim3(256), 0, (hipStream_t)stream, (const f16*)proj, (const f16*)dy, (f16*)dproj, n4);
    return fd_check_launch("fd_geglu_bwd_interleaved");
}
extern "C" int fd_geglu_fwd(const void* proj, void* y, int M, int F, void* stream) {
    FD_REQUIRE((F & 7) == 0, "fd_geglu_fwd: F %% 8");
    hipLaunchKernelGGL(geglu_fwd_kernel, grid_for((int64_t)M * (F / 8)), dim3(256), 0, (hipStream_t)stream, (const f16*)proj, (f16*)y, (int64_t)M, F);
    return fd_check_launch("fd_geglu_fwd");
}
extern "C" int fd_geglu_bwd(const void* proj, const void* dy, void* dproj, int M, int F, void* stream) {
    FD_REQUIRE((F & 7) == 0, "fd_geglu_bwd: F %% 8");
    hipLaunchKernelGGL(geglu_bwd_kernel, grid_for((int64_t)M * (F / 8)), dim3(256), 0, (hipStream_t)stream, (const f16*)proj, (const f16*)dy,
                       (f16*)dproj, (int64_t)M, F);
    return fd_check_launch("fd_geglu_bwd");
}

// ---------------------------------------------------------------- activations / adds / casts (n % 8 == 0 fast path + tail)
__global__ void act_fwd_kernel(const f16* x, f16* y, int64_t n, int act) {
    const int64_t nv = n >> 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        const f16x8 v = *(const f16x8*)(x + i * 8);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)apply_act((float)v[j], act);
        *(f16x8*)(y + i * 8) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const int64_t i = (nv << 3) + threadIdx.x;
        y[i] = (f16)apply_act((float)x[i], act);
    }
}
__global__ void act_bwd_kernel(const f16* z, const f16* dy, f16* dx, int64_t n, int act) {
    const int64_t nv = n >> 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        const f16x8 v = *(const f16x8*)(z + i * 8);
        const f16x8 d = *(const f16x8*)(dy + i * 8);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)((float)d[j] * act_grad((float)v[j], act));
        *(f16x8*)(dx + i * 8) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const int64_t i = (nv << 3) + threadIdx.x;
        dx[i] = (f16)((float)dy[i] * act_grad((float)z[i], act));
    }
}
__global__ void add_kernel(const f16* a, const f16* b, f16* y, int64_t n, float sa, float sb) {
    FD_WG_TRACE(20);
    const int64_t nv = n >> 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        const f16x8 u = *(const f16x8*)(a + i * 8);
        f16x8 w = {0, 0, 0, 0, 0, 0, 0, 0};
        if (b) w = *(const f16x8*)(b + i * 8);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)(sa * (float)u[j] + sb * (float)w[j]);
        *(f16x8*)(y + i * 8) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const int64_t i = (nv << 3) + threadIdx.x;
        y[i] = (f16)(sa * (float)a[i] + (b ? sb * (float)b[i] : 0.f));
    }
}
__global__ void cast_f32_f16_kernel(const float* x, f16* y, int64_t n, float scale) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = (f16)(x[i] * scale);
}
__global__ void cast_f16_f32_kernel(const f16* x, float* y, int64_t n, float scale) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = (float)x[i] * scale;
}
extern "C" int fd_act_fwd(const void* x, void* y, int64_t n, int act, void* stream) {
    hipLaunchKernelGGL(act_fwd_kernel, grid_for(n / 8 + 1), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, n, act);
    return fd_check_launch("fd_act_fwd");
}
extern "C" int fd_act_bwd(const void* z, const void* dy, void* dx, int64_t n, int act, void* stream) {
    hipLaunchKernelGGL(act_bwd_kernel, grid_for(n / 8 + 1), dim3(256), 0, (hipStream_t)stream, (const f16*)z, (const f16*)dy, (f16*)dx, n, act);
    return fd_check_launch("fd_act_bwd");
}
extern "C" int fd_add(const void* a, const void* b, void* y, int64_t n, float sa, float sb, void* stream) {
    hipLaunchKernelGGL(add_kernel, grid_for(n / 8 + 1), dim3(256), 0, (hipStream_t)stream, (const f16*)a, (const f16*)b, (f16*)y, n, sa, sb);
    return fd_check_launch("fd_add");
}
extern "C" int fd_cast_f32_to_f16(const float* x, void* y, int64_t n, float scale, void* stream) {
    hipLaunchKernelGGL(cast_f32_f16_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, x, (f16*)y, n, scale);
    return fd_check_launch("fd_cast_f32_to_f16");
}
extern "C" int fd_cast_f16_to_f32(const void* x, float* y, int64_t n, float scale, void* stream) {
    hipLaunchKernelGGL(cast_f16_f32_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, (const f16*)x, y, n, scale);
    return fd_check_launch("fd_cast_f16_to_f32");
}

// ---------------------------------------------------------------- strided 2-D copy (channel-concat halves)
__global__ void copy_cols_kernel(const f16* src, int64_t lds, f16* dst, int64_t ldd, int64_t M, int cols) {
    const int CV = cols >> 3;
    const int64_t n = M * CV;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / CV;
        const int c = (int)(i - m * CV) * 8;
        *(f16x8*)(dst + m * ldd + c) = *(const f16x8*)(src + m * lds + c);
    }
}
extern "C" int fd_copy_cols(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t M, int cols, void* stream) {
    FD_REQUIRE((cols & 7) == 0 && (lds & 7) == 0 && (ldd & 7) == 0, "fd_copy_cols: multiples of 8");
    hipLaunchKernelGGL(copy_cols_kernel, grid_for(M * (cols / 8)), dim3(256), 0, (hipStream_t)stream, (const f16*)src, lds, (f16*)dst, ldd, M, cols);
    return fd_check_launch("fd_copy_cols");
}

// ---------------------------------------------------------------- [B,T,C] -> [B,C,Tp] (zero-padded keys), 64x64 LDS tiles
__global__ __launch_bounds__(256) void transpose_btc_kernel(const f16* x, f16* y, int T, int C, int Tp, int64_t ldx) {
    __shared__ f16 tile[64][66];
    const int b = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch = tid + i * 256;  // 512 chunks: row = ch/8 (t), col chunk = ch%8
        const int tr = ch >> 3, cc = (ch & 7) * 8;
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t0 + tr < T && c0 + cc < C) v = *(const f16x8*)(x + ((int64_t)b * T + t0 + tr) * ldx + c0 + cc);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[tr][cc + j] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch = tid + i * 256;
        const int cr = ch >> 3, tc = (ch & 7) * 8;
        if (c0 + cr < C && t0 + tc < Tp) {
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = tile[tc + j][cr];
            *(f16x8*)(y + ((int64_t)b * C + c0 + cr) * Tp + t0 + tc) = o;
        }
    }
}
extern "C" int fd_transpose_btc(const void* x, int64_t ldx, void* y, int B, int T, int C, int Tp, void* stream) {
    if (ldx <= 0) ldx = C;
    FD_REQUIRE((C & 7) == 0 && (Tp & 7) == 0 && Tp >= T && (ldx & 7) == 0, "fd_transpose_btc: C%%8, Tp%%8, ldx%%8, Tp>=T");
    dim3 grid((Tp + 63) / 64, (C + 63) / 64, B);
    hipLaunchKernelGGL(transpose_btc_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, T, C, Tp, ldx);
    return fd_check_launch("fd_transpose_btc");
}

// ---------------------------------------------------------------- 2x2 sum (backward of nearest upsample)
__global__ void downsum_kernel(const f16* x, f16* y, int B, int H, int W, int C) {
    const int CV = C >> 3;
    const int64_t n = (int64_t)B * H * W * CV;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * 8;
        int64_t p = i / CV;
        const int xo = (int)(p % W); p /= W;
        const int yo = (int)(p % H);
        const int b = (int)(p / H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const f16x8 v = *(const f16x8*)(x + (((int64_t)b * 2 * H + 2 * yo + dy) * 2 * W + 2 * xo + dx) * C + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
            }
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)acc[j];
        *(f16x8*)(y + (((int64_t)b * H + yo) * W + xo) * C + c) = o;
    }
}

extern "C" int fd_downsum2x2(const void* x, void* y, int B, int H, int W, int C, void* stream) {
    FD_REQUIRE((C & 7) == 0, "fd_downsum2x2: C%%8");
    hipLaunchKernelGGL(downsum_kernel, grid_for((int64_t)B * H * W * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, B, H, W, C);
    return fd_check_launch("fd_downsum2x2");
}

// ---------------------------------------------------------------- row softmax (VAE / CLIP attention), one block per row
// y = softmax(scale*x + mask);  mask row = (row / mask_ht) * mask_t + (row % mask_t)   (fp32, or NULL)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const f16* x, f16* y, int cols, float scale, const float* mask, int mask_t,
                                                           int mask_ht) {
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const f16* xr = x + row * cols;
    const float* mr = mask ? mask + ((row / mask_ht) * mask_t + (row % mask_t)) * (int64_t)cols : nullptr;
    float v[16];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = threadIdx.x + i * 256;
        v[i] = -INFINITY;
        if (c < cols) {
            v[i] = (float)xr[c] * scale + (mr ? mr[c] : 0.f);
            mx = fmaxf(mx, v[i]);
        }
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < cols) {
            v[i] = __expf(v[i] - mx);
            s += v[i];
        }
    }
    s = block_sum(s, red + 4);
    const float inv = 1.f / s;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < cols) y[row * cols + c] = (f16)(v[i] * inv);
    }
}
// ds = scale * p * (dp - sum(p*dp))
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const f16* p, const f16* dp, f16* ds, int cols, float scale) {
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    float pv[16], dv[16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = threadIdx.x + i * 256;
        pv[i] = dv[i] = 0.f;
        if (c < cols) {
            pv[i] = (float)p[row * cols + c];
            dv[i] = (float)dp[row * cols + c];
            s += pv[i] * dv[i];
        }
    }
    s = block_sum(s, red);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < cols) ds[row * cols + c] = (f16)(scale * pv[i] * (dv[i] - s));
    }
}
extern "C" int fd_softmax_rows(const void* x, void* y, int64_t rows, int cols, float scale, const float* mask, int mask_t, int mask_ht,
                               void* stream) {
    FD_REQUIRE(cols > 0 && cols <= 4096 && rows > 0, "fd_softmax_rows: cols must be in 1..4096");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, cols, scale, mask,
                       mask_t > 0 ? mask_t : 1, mask_ht > 0 ? mask_ht : 1);
    return fd_check_launch("fd_softmax_rows");
}
extern "C" int fd_softmax_rows_bwd(const void* p, const void* dp, void* ds, int64_t rows, int cols, float scale, void* stream) {
    FD_REQUIRE(cols > 0 && cols <= 4096 && rows > 0, "fd_softmax_rows_bwd: cols must be in 1..4096");
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const f16*)p, (const f16*)dp, (f16*)ds,
                       cols, scale);
    return fd_check_launch("fd_softmax_rows_bwd");
}

// ---------------------------------------------------------------- layout: channels-last fp16 -> NCHW (fp32 or fp16), optional clamp
__global__ void nhwc_to_nchw_kernel(const f16* x, int64_t ldx, float* y32, f16* y16, int HW, int C, float scale, float lo, float hi, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int hw = (int)(i % HW);
        const int64_t bc = i / HW;
        const int c = (int)(bc % C);
        const int64_t b = bc / C;
        float v = (float)x[(b * HW + hw) * ldx + c] * scale;
        v = fminf(fmaxf(v, lo), hi);
        if (y32) y32[i] = v;
        else y16[i] = (f16)v;
    }
}
extern "C" int fd_nhwc_to_nchw(const void* x, int64_t ldx, void* y, int y_is_f32, int B, int HW, int C, float scale, float lo, float hi,
                               void* stream) {
    const int64_t n = (int64_t)B * HW * C;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, (const f16*)x, ldx, y_is_f32 ? (float*)y : nullptr,
                       y_is_f32 ? nullptr : (f16*)y, HW, C, scale, lo, hi, n);
    return fd_check_launch("fd_nhwc_to_nchw");
}
// dpre[B,C,HW] fp32 = (lo <= pre <= hi) ? dimg : 0, pre channels-last fp16 (backward of clamp(-1,1))
__global__ void clamp_bwd_kernel(const f16* pre, int64_t ldx, const float* dimg, float* dpre, int HW, int C, float lo, float hi, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int hw = (int)(i % HW);
        const int64_t bc = i / HW;
        const int c = (int)(bc % C);
        const int64_t b = bc / C;
        const float v = (float)pre[(b * HW + hw) * ldx + c];
        dpre[i] = (v >= lo && v <= hi) ? dimg[i] : 0.f;
    }
}
extern "C" int fd_clamp_bwd(const void* pre, int64_t ldx, const float* dimg, float* dpre, int B, int HW, int C, float lo, float hi, void* stream) {
    const int64_t n = (int64_t)B * HW * C;
    hipLaunchKernelGGL(clamp_bwd_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, (const f16*)pre, ldx, dimg, dpre, HW, C, lo, hi, n);
    return fd_check_launch("fd_clamp_bwd");
}

// ---------------------------------------------------------------- CFG combine + DPM-Solver++(2M) update
__global__ void cfg_dpm_kernel(const float* eps, float g, float* lat, const float* x0_prev, float* x0_out, float alpha_t, float sigma_t,
                               float c_x, float c_d0, float c_d1, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float eu = eps[i], ec = eps[i + n];
        const float e = eu + g * (ec - eu);
        const float x = lat[i];
        const float x0 = (x - sigma_t * e) / alpha_t;
        float nx = c_x * x - c_d0 * x0;
        if (x0_prev) nx -= c_d1 * (x0 - x0_prev[i]);
        x0_out[i] = x0;
        lat[i] = nx;
    }
}
extern "C" int fd_cfg_dpm_step(const float* eps, float guidance, float* lat, const float* x0_prev, float* x0_out, float alpha_t, float sigma_t,
                               float c_x, float c_d0, float c_d1, int64_t n, void* stream) {
    hipLaunchKernelGGL(cfg_dpm_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, eps, guidance, lat, x0_prev, x0_out, alpha_t, sigma_t, c_x,
                       c_d0, c_d1, n);
    return fd_check_launch("fd_cfg_dpm_step");
}

// ---------------------------------------------------------------- optimizer
__global__ void grad_finite_scale_kernel(float* g, int64_t n, float scale, int32_t* flag) {
    bool bad = false;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = g[i];
        bad |= !isfinite(v);
        g[i] = v * scale;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
extern "C" int fd_grad_finite_scale(float* g, int64_t n, float scale, int32_t* nonfinite_flag, void* stream) {
    hipLaunchKernelGGL(grad_finite_scale_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, g, n, scale, nonfinite_flag);
    return fd_check_launch("fd_grad_finite_scale");
}
// torch.optim.AdamW (decoupled decay, bias-corrected) followed by the EMA update s -= omd*(s-p)
__global__ void adamw_ema_kernel(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr, float b1, float b2, float eps,
                                 float wd, float bc1, float bc2_sqrt, float omd) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pi = p[i] * (1.f - lr * wd);
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);
        const float vi = v[i] * b2 + gi * gi * (1.f - b2);
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (ema) ema[i] -= omd * (ema[i] - pi);
    }
}
extern "C" int fd_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr, float beta1, float beta2, float eps,
                            float weight_decay, int32_t step, float ema_one_minus_decay, void* stream) {
    FD_REQUIRE(step >= 1, "fd_adamw_ema: step counts from 1");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_ema_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, n, lr, beta1, beta2, eps, weight_decay, bc1,
                       bc2s, ema_one_minus_decay);
    return fd_check_launch("fd_adamw_ema");
}

// ---------------------------------------------------------------- ViT patch embedding input (image regularisers, 1-main-debias.py:1139-1175)
// chips [N,3,S,S] fp16 NCHW in [-1,1]  ->  patches [N*g*g, Kp] fp16, k = c*P*P + py*P + px (the Conv2d(3,D,P,P) weight order),
// value ((x+1)/2 - mean_c)/std_c, columns >= 3*P*P zero.
struct PatchNorm { float mean[3], istd[3]; };
__global__ void patchify_fwd_kernel(const f16* chips, f16* patches, PatchNorm nm, int S, int P, int Kp, int64_t n) {
    const int g = S / P, PP = P * P;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp);
        const int64_t row = i / Kp;
        float v = 0.f;
        if (k < 3 * PP) {
            const int c = k / PP, py = (k % PP) / P, px = k % P;
            const int gx = (int)(row % g), gy = (int)((row / g) % g);
            const int64_t b = row / (g * g);
            const float x = (float)chips[((b * 3 + c) * S + gy * P + py) * S + gx * P + px];
            v = ((x + 1.f) * 0.5f - nm.mean[c]) * nm.istd[c];
        }
        patches[i] = (f16)v;
    }
}
// dchips[n,c,y,x] (+)= 0.5/std_c * scale * dpatches[row, k]
__global__ void patchify_bwd_kernel(const f16* dpatches, float* dchips, PatchNorm nm, int S, int P, int Kp, float scale, int accumulate, int64_t n) {
    const int g = S / P, PP = P * P;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % S), y = (int)((i / S) % S), c = (int)((i / ((int64_t)S * S)) % 3);
        const int64_t b = i / ((int64_t)3 * S * S);
        const int64_t row = (b * g + y / P) * g + x / P;
        const float v = (float)dpatches[row * Kp + c * PP + (y % P) * P + x % P] * (0.5f * nm.istd[c] * scale);
        dchips[i] = accumulate ? dchips[i] + v : v;
    }
}
extern "C" int fd_patchify_fwd(const void* chips, void* patches, const float* mean3, const float* std3, int N, int S, int P, int Kp, void* stream) {
    FD_REQUIRE(N > 0 && P > 0 && S % P == 0 && Kp >= 3 * P * P && (Kp & 7) == 0, "fd_patchify_fwd: S %% P, Kp >= 3*P*P, Kp %% 8 (S=%d P=%d Kp=%d)", S, P, Kp);
    PatchNorm nm;
    for (int c = 0; c < 3; ++c) { nm.mean[c] = mean3[c]; nm.istd[c] = 1.f / std3[c]; }
    const int64_t n = (int64_t)N * (S / P) * (S / P) * Kp;
    hipLaunchKernelGGL(patchify_fwd_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, (const f16*)chips, (f16*)patches, nm, S, P, Kp, n);
    return fd_check_launch("fd_patchify_fwd");
}
extern "C" int fd_patchify_bwd(const void* dpatches, float* dchips, const float* std3, int N, int S, int P, int Kp, float scale, int accumulate,
                               void* stream) {
    FD_REQUIRE(N > 0 && P > 0 && S % P == 0 && Kp >= 3 * P * P && (Kp & 7) == 0, "fd_patchify_bwd: S %% P, Kp >= 3*P*P, Kp %% 8");
    PatchNorm nm;
    for (int c = 0; c < 3; ++c) { nm.mean[c] = 0.f; nm.istd[c] = 1.f / std3[c]; }
    const int64_t n = (int64_t)N * 3 * S * S;
    hipLaunchKernelGGL(patchify_bwd_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, (const f16*)dpatches, dchips, nm, S, P, Kp, scale, accumulate, n);
    return fd_check_launch("fd_patchify_bwd");
}

// ---------------------------------------------------------------- rectangle scale of an image gradient (apply_grad_hook_face, :1584-1617)
// dimg [B,3,H,W] fp32: inside rect_b = [x0,y0,x1,y1) multiply by factor_b.  The reference indexes image[:, y0:y1, x0:x1].
__global__ void rect_scale_kernel(float* dimg, const int32_t* rects, const float* factors, int H, int W, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int64_t b = i / ((int64_t)3 * H * W);
        const int32_t* r = rects + b * 4;
        if (x >= r[0] && x < r[2] && y >= r[1] && y < r[3]) dimg[i] *= factors[b];
    }
}
extern "C" int fd_rect_scale(float* dimg, const int32_t* rects, const float* factors, int B, int H, int W, void* stream) {
    const int64_t n = (int64_t)B * 3 * H * W;
    hipLaunchKernelGGL(rect_scale_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, dimg, rects, factors, H, W, n);
    return fd_check_launch("fd_rect_scale");
}


// ---------------------------------------------------------------- fixed-order sum of fp32 slabs (deterministic shared dK / dV, attn.hip)
__global__ void sum_slabs_kernel(const float* __restrict__ in, float* __restrict__ out, int nslab, int64_t n4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 a = ((const f32x4*)in)[i];
        for (int s = 1; s < nslab; ++s) {
            const f32x4 b = ((const f32x4*)in)[(int64_t)s * n4 + i];
            a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
        }
        ((f32x4*)out)[i] = a;
    }
}
extern "C" int fd_sum_slabs(const float* in, float* out, int nslab, int64_t n, void* stream) {
    FD_REQUIRE(in && out && nslab >= 1 && n > 0 && (n & 3) == 0, "fd_sum_slabs: n must be a positive multiple of 4");
    const int64_t n4 = n >> 2;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, nslab, n4);
    return fd_check_launch("fd_sum_slabs");
}

FD_WGT_SETTER(elementwise)
