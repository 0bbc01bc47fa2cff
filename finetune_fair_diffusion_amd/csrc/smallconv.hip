// Direct (VALU) kernels for the thin ends of the path: tiny-channel convolutions (latent/pixel
// side of the U-Net and VAE, classifier stem), depthwise convolutions + squeeze-excite pieces of the
// MobileNetV3 face-attribute classifier, and the differentiable crop+bilinear-resize of the face chip.
#include "common.h"

static inline dim3 grid1d(int64_t n, int threads = 256) {
    int64_t b = (n + threads - 1) / threads;
    if (b > 65535) b = 65535;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}

// ------------------------------------------------------------------ conv with tiny Cin (<= 8), k in {1,3}, pad (k-1)/2
// x: [B,Cin,H,W] (nchw) or [B,H,W,Cin]; fp32 or fp16.  w: fp32 [k*k*Cin][Cout].  y: fp16 [B,Ho,Wo,Cout]
template <typename XT>
__global__ void conv_small_cin_kernel(const XT* x, int nchw, const float* w, const float* bias, f16* y, int B, int H, int W, int Cin, int Cout,
                                      int k, int stride, int Ho, int Wo, int act) {
    const int64_t n = (int64_t)B * Ho * Wo * Cout;
    const int pad = (k - 1) / 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        int64_t p = i / Cout;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float acc = bias ? bias[co] : 0.f;
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * stride + ky - pad;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * stride + kx - pad;
                if (ix < 0 || ix >= W) continue;
                for (int ci = 0; ci < Cin; ++ci) {
                    const float xv = nchw ? (float)x[(((int64_t)b * Cin + ci) * H + iy) * W + ix] : (float)x[(((int64_t)b * H + iy) * W + ix) * Cin + ci];
                    acc += xv * w[((ky * k + kx) * Cin + ci) * Cout + co];
                }
            }
        }
        y[i] = (f16)apply_act(acc, act);
    }
}
// Fast path for the shapes on the hot path (U-Net conv_in 4 -> 320 and the data gradient of conv_out as a 4 -> 320 conv, both at 64^2 x 16
// samples; VAE conv_in 4 -> 512; classifier stem 3 -> 16 stride 2): k = 3, Cin in {3, 4}, Cout % 8 == 0.  The kernel above spends one thread per
// output ELEMENT (36 input + 36 weight loads for one 2-byte store: 344 us for 42 MB of output); here a thread owns 8 output channels of PX = 4
// neighbouring output pixels: the 3 x (3*STRIDE + 3) x Cin input patch is loaded once into registers (the CG threads of a pixel group read the
// same addresses: one transaction), the weights [9*Cin][Cout] sit in LDS (two 16-byte reads per tap feed 32 FMAs), the store is 16 bytes
// per lane over whole 2*Cout-byte pixel rows.  Same summation order per output as the kernel above (bias, then ky, kx, ci).
template <typename XT, int CIN, int STRIDE, bool NCHW, bool HAS_ACT>
__global__ __launch_bounds__(256, 2) void conv_small_cin_fast_kernel(const XT* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ bias, f16* __restrict__ y, int B, int H, int W, int Cout,
                                                                   int Ho, int Wo, int act) {
    constexpr int K = 3, PX = 4, COLS = (PX - 1) * STRIDE + K;
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [9*CIN][Cout]
    for (int i = threadIdx.x; i < K * K * CIN * Cout / 4; i += 256) ((f32x4*)wl)[i] = ((const f32x4*)w)[i];
    __syncthreads();
    const int CG = Cout >> 3, slots = 256 / CG;
    const int slot = threadIdx.x / CG, cg = threadIdx.x - slot * CG;
    if (slot >= slots) return;
    const int c0 = cg * 8;
    const int qw = (Wo + PX - 1) / PX;
    const int64_t nq = (int64_t)B * Ho * qw;
    float bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = bias ? bias[c0 + j] : 0.f;
#pragma unroll 1
    for (int64_t q = (int64_t)blockIdx.x * slots + slot; q < nq; q += (int64_t)gridDim.x * slots) {
        const int qx = (int)(q % qw);
        int64_t p = q / qw;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const int ox0 = qx * PX;
        const int iy0 = oy * STRIDE - 1, ix0 = ox0 * STRIDE - 1;
        float acc[PX][8];
#pragma unroll
        for (int px = 0; px < PX; ++px)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[px][j] = bv[j];
        // the weight reads of a quad stay LDS reads next to their use: left visible as loop-invariant the compiler hoists all 288 weights
        // into registers (395 VGPR+AGPR, one wave per SIMD -- or scratch under an occupancy target)
        int wofs = c0;
        asm volatile("" : "+v"(wofs));
        // one input row (COLS x CIN values) live at a time, the next row's loads issued before this row's FMAs
        float xr[COLS][CIN], xn[COLS][CIN];
        const XT* xb = x + (int64_t)b * CIN * H * W;     // this image; everything below is 32-bit offsets from it
        auto load_row = [&](int r, float (&dst)[COLS][CIN]) {
            const int iy = iy0 + r;
            const bool rok = iy >= 0 && iy < H;
            const int rowoff = (rok ? iy : 0) * W;
#pragma unroll
            for (int c = 0; c < COLS; ++c) {
                const int ix = ix0 + c;
                const bool ok = rok && ix >= 0 && ix < W;
                const int po = rowoff + (ok ? ix : 0);
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    const float v = NCHW ? (float)xb[ci * H * W + po] : (float)xb[po * CIN + ci];
                    dst[c][ci] = ok ? v : 0.f;
                }
            }
        };
        load_row(0, xr);
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            if (ky + 1 < K) load_row(ky + 1, xn);
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    const float* wp = wl + ((ky * K + kx) * CIN + ci) * Cout + wofs;
                    const f32x4 w0 = *(const f32x4*)wp, w1 = *(const f32x4*)(wp + 4);
#pragma unroll
                    for (int px = 0; px < PX; ++px) {
                        const float xv = xr[px * STRIDE + kx][ci];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[px][j] += xv * w0[j];
                            acc[px][4 + j] += xv * w1[j];
                        }
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < COLS; ++c)
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) xr[c][ci] = xn[c][ci];
        }
        // (the rolled loop below also keeps register allocation sane: an instantiation compiled WITHOUT it spilled 900 B per lane at
        // 256 VGPRs against 120-156 VGPRs and no scratch with it, so there is only this one form; act == none skips it at run time)
        if (HAS_ACT && act != FD_ACT_NONE) {
#pragma unroll 1
            for (int px = 0; px < PX; ++px)       // rolled: the generic activation switch is large, it must not be expanded 32 times
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[px][j] = apply_act(acc[px][j], act);
        }
#pragma unroll
        for (int px = 0; px < PX; ++px) {
            if (ox0 + px < Wo) {
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (f16)acc[px][j];
                *(f16x8*)(y + (((int64_t)b * Ho + oy) * Wo + ox0 + px) * Cout + c0) = o;
            }
        }
    }
}

template <typename XT, int CIN, int STRIDE, bool NCHW, bool HAS_ACT>
static void launch_conv_small_fast_a(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int Cout, int Ho, int Wo,
                                   int act, hipStream_t s) {
    const size_t lds = (size_t)9 * CIN * Cout * sizeof(float);
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)conv_small_cin_fast_kernel<XT, CIN, STRIDE, NCHW, HAS_ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    const int slots = 256 / (Cout / 8);
    const int64_t nq = (int64_t)B * Ho * ((Wo + 3) / 4);
    int64_t blocks = (nq + slots - 1) / slots;
    if (blocks > 512) blocks = 512;           // two workgroups per CU; each first stages the weights (46 KB at Cout = 320): with one pass of quads
                                              // per workgroup that staging was most of the kernel (145 us for the U-Net conv_in at 1366 workgroups)
    hipLaunchKernelGGL((conv_small_cin_fast_kernel<XT, CIN, STRIDE, NCHW, HAS_ACT>), dim3((unsigned)blocks), dim3(256), lds, s, (const XT*)x, w, bias, (f16*)y, B,
                       H, W, Cout, Ho, Wo, act);
}

template <typename XT, int CIN, int STRIDE, bool NCHW>
static void launch_conv_small_fast(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int Cout, int Ho, int Wo, int act,
                                   hipStream_t s) {
    launch_conv_small_fast_a<XT, CIN, STRIDE, NCHW, true>(x, w, bias, y, B, H, W, Cout, Ho, Wo, act, s);
}

extern "C" int fd_conv_small_cin(const void* x, int x_is_f32, int nchw, const float* w, const float* bias, void* y, int B, int H, int W, int Cin,
                                 int Cout, int ksize, int stride, int act, void* stream) {
    FD_REQUIRE(Cin >= 1 && Cin <= 8 && (ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), "fd_conv_small_cin: Cin<=8, k in {1,3}");
    const int pad = (ksize - 1) / 2;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const int64_t n = (int64_t)B * Ho * Wo * Cout;
    hipStream_t s = (hipStream_t)stream;
    if (ksize == 3 && (Cin == 3 || Cin == 4) && (Cout & 7) == 0 && Cout >= 16 && Cout <= 2048 && (size_t)9 * Cin * Cout * 4 <= 150 * 1024 &&
        ((uintptr_t)w & 15) == 0 && (int64_t)B * Cin * H * W < (1LL << 31)) {
#define FD_FAST(XT, CI, ST)                                                                                  \
    do {                                                                                                     \
        if (nchw) launch_conv_small_fast<XT, CI, ST, true>(x, w, bias, y, B, H, W, Cout, Ho, Wo, act, s);    \
        else launch_conv_small_fast<XT, CI, ST, false>(x, w, bias, y, B, H, W, Cout, Ho, Wo, act, s);        \
    } while (0)
        if (x_is_f32) {
            if (Cin == 4) { if (stride == 1) FD_FAST(float, 4, 1); else FD_FAST(float, 4, 2); }
            else { if (stride == 1) FD_FAST(float, 3, 1); else FD_FAST(float, 3, 2); }
        } else {
            if (Cin == 4) { if (stride == 1) FD_FAST(f16, 4, 1); else FD_FAST(f16, 4, 2); }
            else { if (stride == 1) FD_FAST(f16, 3, 1); else FD_FAST(f16, 3, 2); }
        }
#undef FD_FAST
        return fd_check_launch("fd_conv_small_cin(fast)");
    }
    if (x_is_f32)
        hipLaunchKernelGGL(conv_small_cin_kernel<float>, grid1d(n), dim3(256), 0, s, (const float*)x, nchw, w, bias, (f16*)y, B, H,
                           W, Cin, Cout, ksize, stride, Ho, Wo, act);
    else
        hipLaunchKernelGGL(conv_small_cin_kernel<f16>, grid1d(n), dim3(256), 0, s, (const f16*)x, nchw, w, bias, (f16*)y, B, H, W,
                           Cin, Cout, ksize, stride, Ho, Wo, act);
    return fd_check_launch("fd_conv_small_cin");
}

// data gradient of the above: dx[B,Cin,H,W] fp32 NCHW (overwritten) from dy [B,Ho,Wo,Cout] fp16
__global__ void conv_small_cin_bwd_kernel(const f16* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Cout, int k, int stride,
                                          int Ho, int Wo, float scale) {
    const int64_t n = (int64_t)B * H * W;
    const int pad = (k - 1) / 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W);
        int64_t p = i / W;
        const int yy = (int)(p % H);
        const int b = (int)(p / H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int ky = 0; ky < k; ++ky) {
            const int ty = yy + pad - ky;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int tx = xx + pad - kx;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                const f16* g = dy + (((int64_t)b * Ho + oy) * Wo + ox) * Cout;
                for (int co = 0; co < Cout; ++co) {
                    const float gv = (float)g[co];
                    for (int ci = 0; ci < Cin; ++ci) acc[ci] += gv * w[((ky * k + kx) * Cin + ci) * Cout + co];
                }
            }
        }
        for (int ci = 0; ci < Cin; ++ci) dx[(((int64_t)b * Cin + ci) * H + yy) * W + xx] = acc[ci] * scale;
    }
}
extern "C" int fd_conv_small_cin_bwd(const void* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                                     float scale, void* stream) {
    FD_REQUIRE(Cin >= 1 && Cin <= 8 && (ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), "fd_conv_small_cin_bwd: Cin<=8, k in {1,3}");
    const int pad = (ksize - 1) / 2;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    hipLaunchKernelGGL(conv_small_cin_bwd_kernel, grid1d((int64_t)B * H * W), dim3(256), 0, (hipStream_t)stream, (const f16*)dy, w, dx, B, H, W, Cin,
                       Cout, ksize, stride, Ho, Wo, scale);
    return fd_check_launch("fd_conv_small_cin_bwd");
}

// ------------------------------------------------------------------ depthwise conv (channels-last, 8 channels per thread)
// x [B,H,W,C] fp16, w fp32 [k*k][C], bias fp32 [C] -> y [B,Ho,Wo,C]; pad (k-1)/2
__global__ void dwconv_fwd_kernel(const f16* x, const float* w, const float* bias, f16* y, int B, int H, int W, int C, int k, int stride, int Ho,
                                  int Wo, int act) {
    const int CV = C >> 3;
    const int64_t n = (int64_t)B * Ho * Wo * CV;
    const int pad = (k - 1) / 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * 8;
        int64_t p = i / CV;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = bias ? bias[c + j] : 0.f;
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * stride + ky - pad;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * stride + kx - pad;
                if (ix < 0 || ix >= W) continue;
                const f16x8 xv = *(const f16x8*)(x + (((int64_t)b * H + iy) * W + ix) * C + c);
                const float* wp = w + (ky * k + kx) * C + c;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)xv[j] * wp[j];
            }
        }
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)apply_act(acc[j], act);
        *(f16x8*)(y + i * 8) = o;
    }
}
__global__ void dwconv_bwd_kernel(const f16* dy, const float* w, f16* dx, int B, int H, int W, int C, int k, int stride, int Ho, int Wo) {
    const int CV = C >> 3;
    const int64_t n = (int64_t)B * H * W * CV;
    const int pad = (k - 1) / 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * 8;
        int64_t p = i / CV;
        const int xx = (int)(p % W); p /= W;
        const int yy = (int)(p % H);
        const int b = (int)(p / H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int ky = 0; ky < k; ++ky) {
            const int ty = yy + pad - ky;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int tx = xx + pad - kx;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                const f16x8 g = *(const f16x8*)(dy + (((int64_t)b * Ho + oy) * Wo + ox) * C + c);
                const float* wp = w + (ky * k + kx) * C + c;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)g[j] * wp[j];
            }
        }
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)acc[j];
        *(f16x8*)(dx + i * 8) = o;
    }
}
extern "C" int fd_dwconv_fwd(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int k, int stride, int act,
                             void* stream) {
    FD_REQUIRE((C & 7) == 0 && (k == 3 || k == 5) && (stride == 1 || stride == 2), "fd_dwconv_fwd: C%%8, k in {3,5}");
    const int pad = (k - 1) / 2, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    hipLaunchKernelGGL(dwconv_fwd_kernel, grid1d((int64_t)B * Ho * Wo * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, w, bias, (f16*)y,
                       B, H, W, C, k, stride, Ho, Wo, act);
    return fd_check_launch("fd_dwconv_fwd");
}
extern "C" int fd_dwconv_bwd(const void* dy, const float* w, void* dx, int B, int H, int W, int C, int k, int stride, void* stream) {
    FD_REQUIRE((C & 7) == 0 && (k == 3 || k == 5) && (stride == 1 || stride == 2), "fd_dwconv_bwd: C%%8, k in {3,5}");
    const int pad = (k - 1) / 2, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    hipLaunchKernelGGL(dwconv_bwd_kernel, grid1d((int64_t)B * H * W * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const f16*)dy, w, (f16*)dx, B, H,
                       W, C, k, stride, Ho, Wo);
    return fd_check_launch("fd_dwconv_bwd");
}

// ------------------------------------------------------------------ global average pool over HW and its backward; SE channel scaling
__global__ void avgpool_kernel(const f16* x, f16* y, int HW, int C) {  // grid (C/64.., B), block 256 = 64 ch x 4 row-groups
    __shared__ float red[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int r = rg; r < HW; r += 4) s += (float)x[((int64_t)b * HW + r) * C + c];
    red[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && c < C) y[(int64_t)b * C + c] = (f16)((red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / HW);
}
__global__ void avgpool_bwd_kernel(const f16* dy, const f16* add, f16* dx, int HW, int C, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t b = i / ((int64_t)HW * C);
        dx[i] = (f16)((float)dy[b * C + c] / HW + (add ? (float)add[i] : 0.f));
    }
}
__global__ void scale_channels_kernel(const f16* x, const f16* s, f16* y, int HW, int C, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t b = i / ((int64_t)HW * C);
        y[i] = (f16)((float)x[i] * (float)s[b * C + c]);
    }
}
// dx = dy * s ; ds[b,c] = sum_hw dy * x
__global__ void scale_channels_bwd_kernel(const f16* x, const f16* s, const f16* dy, f16* dx, f16* ds, int HW, int C) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < C) {
        const float sv = (float)s[(int64_t)b * C + c];
        for (int r = rg; r < HW; r += 4) {
            const int64_t i = ((int64_t)b * HW + r) * C + c;
            const float g = (float)dy[i];
            acc += g * (float)x[i];
            dx[i] = (f16)(g * sv);
        }
    }
    red[rg][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rg == 0 && c < C) ds[(int64_t)b * C + c] = (f16)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
extern "C" int fd_avgpool_hw(const void* x, void* y, int B, int HW, int C, void* stream) {
    hipLaunchKernelGGL(avgpool_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, HW, C);
    return fd_check_launch("fd_avgpool_hw");
}
extern "C" int fd_avgpool_hw_bwd(const void* dy, const void* add, void* dx, int B, int HW, int C, void* stream) {
    const int64_t n = (int64_t)B * HW * C;
    hipLaunchKernelGGL(avgpool_bwd_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, (const f16*)dy, (const f16*)add, (f16*)dx, HW, C, n);
    return fd_check_launch("fd_avgpool_hw_bwd");
}
extern "C" int fd_scale_channels(const void* x, const void* s, void* y, int B, int HW, int C, void* stream) {
    const int64_t n = (int64_t)B * HW * C;
    hipLaunchKernelGGL(scale_channels_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (const f16*)s, (f16*)y, HW, C, n);
    return fd_check_launch("fd_scale_channels");
}
extern "C" int fd_scale_channels_bwd(const void* x, const void* s, const void* dy, void* dx, void* ds, int B, int HW, int C, void* stream) {
    hipLaunchKernelGGL(scale_channels_bwd_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (const f16*)s,
                       (const f16*)dy, (f16*)dx, (f16*)ds, HW, C);
    return fd_check_launch("fd_scale_channels_bwd");
}

// ------------------------------------------------------------------ crop [x0,y0,x1,y1) + constant pad + bilinear resize to SxS
// (torchvision Pad + Resize on tensors: align_corners=False, no antialias; 1-main-debias.py:267-290)
__device__ __forceinline__ void bilinear_src(int o, int in_size, int out_size, int& i0, int& i1, float& lam) {
    float src = ((float)o + 0.5f) * ((float)in_size / (float)out_size) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    lam = src - (float)i0;
}
// img [B,3,H,W] fp16 -> chips [B,3,S,S] fp16
__global__ void crop_resize_fwd_kernel(const f16* img, const int32_t* boxes, float fill, f16* chips, int H, int W, int S, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % S);
        int64_t p = i / S;
        const int oy = (int)(p % S); p /= S;
        const int c = (int)(p % 3);
        const int b = (int)(p / 3);
        const int x0 = boxes[b * 4], y0 = boxes[b * 4 + 1], x1 = boxes[b * 4 + 2], y1 = boxes[b * 4 + 3];
        int ya, yb, xa, xb;
        float ly, lx;
        bilinear_src(oy, y1 - y0, S, ya, yb, ly);
        bilinear_src(ox, x1 - x0, S, xa, xb, lx);
        const f16* ip = img + ((int64_t)b * 3 + c) * H * W;
        auto at = [&](int py, int px) -> float {
            const int yy = y0 + py, xx = x0 + px;
            return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (float)ip[(int64_t)yy * W + xx] : fill;
        };
        const float v = (1.f - ly) * ((1.f - lx) * at(ya, xa) + lx * at(ya, xb)) + ly * ((1.f - lx) * at(yb, xa) + lx * at(yb, xb));
        chips[i] = (f16)v;
    }
}
// dchips [B,3,S,S] fp32 -> dimg [B,3,H,W] fp32, ADDED to what dimg holds (the caller zeroes it or passes an accumulator).
// Gather form, one thread per image pixel: the chip pixels whose bilinear footprint contains the pixel form a small contiguous range per
// axis (the source coordinate is monotonic in the output index), visited in a fixed order -- the result is bit-reproducible.  The scatter
// form with fp32 atomics this replaces made dL/d(image), and through the VAE backward's one fp16 rounding every LoRA gradient of the step,
// vary from run to run at the fp16-ulp level (1e-3 of max |g| between two identical steps).
__device__ __forceinline__ void bilinear_taps(int p, int in_size, int out_size, int& lo, int& hi) {
    // output indices o whose two source taps can include source index p: src(o) in [p - 1, p + 1)  (+- 1 of slack, filtered exactly later)
    const float r = (float)out_size / (float)in_size;
    lo = (int)floorf(((float)p - 0.5f) * r - 0.5f) - 1;
    hi = (int)ceilf(((float)p + 1.5f) * r - 0.5f) + 1;
    if (lo < 0) lo = 0;
    if (hi > out_size - 1) hi = out_size - 1;
}
__global__ void crop_resize_bwd_kernel(const float* dchips, const int32_t* boxes, float* dimg, int H, int W, int S, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W);
        int64_t p = i / W;
        const int yy = (int)(p % H); p /= H;
        const int c = (int)(p % 3);
        const int b = (int)(p / 3);
        const int x0 = boxes[b * 4], y0 = boxes[b * 4 + 1], x1 = boxes[b * 4 + 2], y1 = boxes[b * 4 + 3];
        const int px = xx - x0, py = yy - y0;
        if (px < 0 || py < 0 || px >= x1 - x0 || py >= y1 - y0) continue;
        int oylo, oyhi, oxlo, oxhi;
        bilinear_taps(py, y1 - y0, S, oylo, oyhi);
        bilinear_taps(px, x1 - x0, S, oxlo, oxhi);
        const float* gp = dchips + ((int64_t)b * 3 + c) * S * S;
        float acc = 0.f;
        for (int oy = oylo; oy <= oyhi; ++oy) {
            int ya, yb;
            float ly;
            bilinear_src(oy, y1 - y0, S, ya, yb, ly);
            const float wy = (ya == py ? 1.f - ly : 0.f) + (yb == py ? ly : 0.f);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int ox = oxlo; ox <= oxhi; ++ox) {
                int xa, xb;
                float lx;
                bilinear_src(ox, x1 - x0, S, xa, xb, lx);
                const float wx = (xa == px ? 1.f - lx : 0.f) + (xb == px ? lx : 0.f);
                row += wx * gp[(int64_t)oy * S + ox];
            }
            acc += wy * row;
        }
        dimg[i] += acc;
    }
}
extern "C" int fd_crop_resize_fwd(const void* img, const int32_t* boxes, float fill, void* chips, int B, int H, int W, int S, void* stream) {
    const int64_t n = (int64_t)B * 3 * S * S;
    hipLaunchKernelGGL(crop_resize_fwd_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, (const f16*)img, boxes, fill, (f16*)chips, H, W, S, n);
    return fd_check_launch("fd_crop_resize_fwd");
}
extern "C" int fd_crop_resize_bwd(const float* dchips, const int32_t* boxes, float* dimg, int B, int H, int W, int S, void* stream) {
    const int64_t n = (int64_t)B * 3 * H * W;       // one thread per IMAGE pixel (gather)
    hipLaunchKernelGGL(crop_resize_bwd_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, dchips, boxes, dimg, H, W, S, n);
    return fd_check_launch("fd_crop_resize_bwd");
}

// ---------------------------------------------------------------- affine face alignment (image_pipeline, 1-main-debias.py:292-312)
// kornia.warp_affine(bilinear, zeros padding) applied to (img+1)/2*255 and mapped back: in [-1,1] space that is bilinear sampling
// with out-of-bounds taps reading -1 (``fill``).  A[n] = 2x3 map from output pixel (x, y, 1) to the sampling position in input
// pixel units (grid_sample convention already folded in on the host).  img [B,3,H,W] fp16, src_index[n] selects the image of chip n.
__global__ void warp_affine_fwd_kernel(const f16* img, const int32_t* src_index, const float* A, float fill, f16* chips, int H, int W,
                                       int S, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % S);
        int64_t p = i / S;
        const int oy = (int)(p % S); p /= S;
        const int c = (int)(p % 3);
        const int k = (int)(p / 3);
        const float* a = A + k * 6;
        const float xs = a[0] * ox + a[1] * oy + a[2], ys = a[3] * ox + a[4] * oy + a[5];
        const float xf = floorf(xs), yf = floorf(ys);
        const int x0 = (int)xf, y0 = (int)yf;
        const float lx = xs - xf, ly = ys - yf;
        const f16* ip = img + ((int64_t)src_index[k] * 3 + c) * H * W;
        auto at = [&](int yy, int xx) -> float { return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (float)ip[(int64_t)yy * W + xx] : fill; };
        const float v = (1.f - ly) * ((1.f - lx) * at(y0, x0) + lx * at(y0, x0 + 1)) + ly * ((1.f - lx) * at(y0 + 1, x0) + lx * at(y0 + 1, x0 + 1));
        chips[i] = (f16)v;
    }
}
// Backward as a fixed-order GATHER (round 4; it was a scatter with fp32 atomics, the last order-free sum on the way to dL/d(image)): one thread
// per IMAGE pixel.  For every chip k that samples this image (ascending k), the output pixels whose bilinear footprint can contain the pixel lie
// in the inverse-mapped bounding box of [xx-1, xx+1] x [yy-1, yy+1]; each candidate recomputes its sampling position with the forward's own
// expression and adds its tap weight, oy-major / ox-minor.  dimg[b,c,yy,xx] += sum -- no two threads touch one element.
__global__ void warp_affine_bwd_kernel(const float* dchips, const int32_t* src_index, const float* A, float* dimg, int n_chips, int H, int W,
                                       int S, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W);
        int64_t p = i / W;
        const int yy = (int)(p % H); p /= H;
        const int c = (int)(p % 3);
        const int b = (int)(p / 3);
        float acc = 0.f;
        bool any = false;
        for (int k = 0; k < n_chips; ++k) {
            if (src_index[k] != b) continue;
            const float* a = A + k * 6;
            const float det = a[0] * a[4] - a[1] * a[3];
            int oxlo = 0, oxhi = S - 1, oylo = 0, oyhi = S - 1;
            if (fabsf(det) > 1e-12f) {
                const float i00 = a[4] / det, i01 = -a[1] / det, i10 = -a[3] / det, i11 = a[0] / det;
                float xlo = 1e30f, xhi = -1e30f, ylo = 1e30f, yhi = -1e30f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float X = (float)(xx + ((q & 1) ? 1 : -1)) - a[2], Y = (float)(yy + ((q & 2) ? 1 : -1)) - a[5];
                    const float ox = i00 * X + i01 * Y, oy = i10 * X + i11 * Y;
                    xlo = fminf(xlo, ox); xhi = fmaxf(xhi, ox); ylo = fminf(ylo, oy); yhi = fmaxf(yhi, oy);
                }
                if (xhi < -1.f || yhi < -1.f || xlo > (float)S || ylo > (float)S) continue;
                oxlo = max(0, (int)floorf(xlo) - 1); oxhi = min(S - 1, (int)ceilf(xhi) + 1);
                oylo = max(0, (int)floorf(ylo) - 1); oyhi = min(S - 1, (int)ceilf(yhi) + 1);
            }
            const float* gp = dchips + ((int64_t)k * 3 + c) * S * S;
            for (int oy = oylo; oy <= oyhi; ++oy)
                for (int ox = oxlo; ox <= oxhi; ++ox) {
                    const float xs = a[0] * ox + a[1] * oy + a[2], ys = a[3] * ox + a[4] * oy + a[5];
                    const float xf = floorf(xs), yf = floorf(ys);
                    const int dx = xx - (int)xf, dy = yy - (int)yf;
                    if (dx < 0 || dx > 1 || dy < 0 || dy > 1) continue;
                    const float lx = xs - xf, ly = ys - yf;
                    acc += gp[(int64_t)oy * S + ox] * ((dy ? ly : 1.f - ly) * (dx ? lx : 1.f - lx));
                    any = true;
                }
        }
        if (any) dimg[i] += acc;
    }
}
extern "C" int fd_warp_affine_fwd(const void* img, const int32_t* src_index, const float* A, float fill, void* chips, int n_chips, int H, int W,
                                  int S, void* stream) {
    const int64_t n = (int64_t)n_chips * 3 * S * S;
    if (n == 0) return FD_OK;
    hipLaunchKernelGGL(warp_affine_fwd_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, (const f16*)img, src_index, A, fill, (f16*)chips, H, W, S, n);
    return fd_check_launch("fd_warp_affine_fwd");
}
extern "C" int fd_warp_affine_bwd(const float* dchips, const int32_t* src_index, const float* A, float* dimg, int n_chips, int B, int H, int W,
                                  int S, void* stream) {
    if (n_chips == 0) return FD_OK;
    const int64_t n = (int64_t)B * 3 * H * W;       // one thread per IMAGE pixel (gather)
    hipLaunchKernelGGL(warp_affine_bwd_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, dchips, src_index, A, dimg, n_chips, H, W, S, n);
    return fd_check_launch("fd_warp_affine_bwd");
}
