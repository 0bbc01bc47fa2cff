// LoRA weight gradients: skinny "TN" reductions over the token dimension.
//   G[n, r] += scale * sum_m X[m, n] * T[m, r]        (X,T fp16; G fp32 with arbitrary strides)
// dUp[N,r]  = dY^T . t   (X = dY, T = t = x.down^T)        dDown[r,K] = u^T . x  (X = x, T = u = dY.up)
// HBM-bound (reads X once); VALU FMAs, coalesced over n, T rows broadcast.  Two-pass with a fixed
// reduction order so repeated runs are bit-identical.
#include "common.h"

template <int RP>
__global__ __launch_bounds__(256) void lora_wgrad_partial(const f16* __restrict__ X, int64_t ldx, const f16* __restrict__ T, int64_t ldt,
                                                          float* __restrict__ partial, int M, int N, int rows_per_split) {
    __shared__ float red[3][64][RP + 1];
    const int nl = threadIdx.x & 63, mg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + nl;
    const int m0 = blockIdx.y * rows_per_split;
    const int m1 = min(m0 + rows_per_split, M);
    float acc[RP];
#pragma unroll
    for (int r = 0; r < RP; ++r) acc[r] = 0.f;
    if (n < N) {
        for (int m = m0 + mg; m < m1; m += 4) {
            const float x = (float)X[(int64_t)m * ldx + n];
            const f16* tr = T + (int64_t)m * ldt;
#pragma unroll
            for (int r8 = 0; r8 < RP; r8 += 8) {
                const f16x8 tv = *(const f16x8*)(tr + r8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[r8 + j] += x * (float)tv[j];
            }
        }
    }
    if (mg > 0) {
#pragma unroll
        for (int r = 0; r < RP; ++r) red[mg - 1][nl][r] = acc[r];
    }
    __syncthreads();
    if (mg == 0 && n < N) {
#pragma unroll
        for (int r = 0; r < RP; ++r) {
            const float v = acc[r] + red[0][nl][r] + red[1][nl][r] + red[2][nl][r];
            partial[((int64_t)blockIdx.y * N + n) * RP + r] = v;
        }
    }
}

// one wave per output element (n, r): lanes stride over the splits in a fixed order, then a fixed-shape wave reduction
__global__ __launch_bounds__(256) void lora_wgrad_final(const float* partial, float* G, int64_t sn, int64_t sr, int N, int R, int RP, int nsplit,
                                                        float scale) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= N * R) return;
    const int lane = threadIdx.x & 63;
    const int n = o / R, r = o % R;
    float s = 0.f;
    for (int k = lane; k < nsplit; k += 64) s += partial[((int64_t)k * N + n) * RP + r];
    s = wave_sum(s);
    if (lane == 0) G[n * sn + r * sr] += scale * s;
}

extern "C" int fd_lora_wgrad(const void* X, int64_t ldx, const void* T, int64_t ldt, float* G, int64_t g_stride_n, int64_t g_stride_r, int M,
                             int N, int R, float scale, float* scratch, int64_t scratch_elems, void* stream) {
    FD_REQUIRE(M > 0 && N > 0 && R > 0 && R <= 64, "fd_lora_wgrad: rank must be in 1..64 (got %d)", R);
    const int RP = R <= 8 ? 8 : (R <= 16 ? 16 : (R <= 32 ? 32 : 64));
    FD_REQUIRE(ldt >= RP && (ldt & 7) == 0, "fd_lora_wgrad: T must be padded to %d columns (ldt=%ld)", RP, (long)ldt);
    const int ncb = (N + 63) / 64;
    int nsplit = (768 + ncb - 1) / ncb;
    if (nsplit > (M + 63) / 64) nsplit = (M + 63) / 64;
    while (nsplit > 1 && (int64_t)nsplit * N * RP > scratch_elems) nsplit >>= 1;
    FD_REQUIRE((int64_t)nsplit * N * RP <= scratch_elems, "fd_lora_wgrad: scratch too small");
    int rows = (M + nsplit - 1) / nsplit;
    rows = (rows + 3) & ~3;
    nsplit = (M + rows - 1) / rows;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(ncb, nsplit);
    switch (RP) {
        case 8: hipLaunchKernelGGL(lora_wgrad_partial<8>, grid, dim3(256), 0, s, (const f16*)X, ldx, (const f16*)T, ldt, scratch, M, N, rows); break;
        case 16: hipLaunchKernelGGL(lora_wgrad_partial<16>, grid, dim3(256), 0, s, (const f16*)X, ldx, (const f16*)T, ldt, scratch, M, N, rows); break;
        case 32: hipLaunchKernelGGL(lora_wgrad_partial<32>, grid, dim3(256), 0, s, (const f16*)X, ldx, (const f16*)T, ldt, scratch, M, N, rows); break;
        default: hipLaunchKernelGGL(lora_wgrad_partial<64>, grid, dim3(256), 0, s, (const f16*)X, ldx, (const f16*)T, ldt, scratch, M, N, rows); break;
    }
    hipLaunchKernelGGL(lora_wgrad_final, dim3((N * R + 3) / 4), dim3(256), 0, s, scratch, G, g_stride_n, g_stride_r, N, R, RP, nsplit, scale);
    return fd_check_launch("fd_lora_wgrad");
}
