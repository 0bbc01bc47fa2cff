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

// Vectorised variant for small ranks: a thread owns CV consecutive columns (16-byte loads of X for CV = 8) and one row phase; a block
// covers CGB column groups x RT = 256 / CGB row phases (CGB = 40: 320 columns, the U-Net's channel granule).  The RT partial sums are
// combined through LDS in a fixed order.  Reads X at full line width instead of 2 bytes per lane.
template <int RP, int CV, int CGB>
__global__ __launch_bounds__(256) void lora_wgrad_partial_vec(const f16* __restrict__ X, int64_t ldx, const f16* __restrict__ T, int64_t ldt,
                                                              float* __restrict__ partial, int M, int N, int rows_per_split) {
    constexpr int RT = 256 / CGB;
    __shared__ float red[(RT - 1) * CGB * (CV * RP + 1)];
    const int cgl = threadIdx.x % CGB, rt = threadIdx.x / CGB;
    const int n0 = (blockIdx.x * CGB + cgl) * CV;
    const int m0 = blockIdx.y * rows_per_split;
    const int m1 = min(m0 + rows_per_split, M);
    float acc[CV][RP];
#pragma unroll
    for (int c = 0; c < CV; ++c)
#pragma unroll
        for (int r = 0; r < RP; ++r) acc[c][r] = 0.f;
    const bool active = rt < RT && n0 < N;      // N % CV == 0 is required by the launcher
    if (active) {
        constexpr int U = 4;                       // rows in flight per thread: the loop is latency-bound without this
        for (int m = m0 + rt; m < m1; m += U * RT) {
            f16 xv[U][CV];
            f16x8 tv[U][RP / 8];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int mm = m + u * RT;
                const bool ok = mm < m1;
                const int64_t mc = ok ? mm : m;   // clamp: the duplicate row is multiplied by zero below
                if (CV == 8) *(f16x8*)xv[u] = *(const f16x8*)(X + mc * ldx + n0);
                else if (CV == 4) *(f16x4*)xv[u] = *(const f16x4*)(X + mc * ldx + n0);
                else
#pragma unroll
                    for (int c = 0; c < CV; ++c) xv[u][c] = X[mc * ldx + n0 + c];
#pragma unroll
                for (int r8 = 0; r8 < RP / 8; ++r8) tv[u][r8] = *(const f16x8*)(T + mc * ldt + r8 * 8);
                if (!ok)
#pragma unroll
                    for (int c = 0; c < CV; ++c) xv[u][c] = (f16)0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float tf[RP];
#pragma unroll
                for (int r = 0; r < RP; ++r) tf[r] = (float)tv[u][r >> 3][r & 7];
#pragma unroll
                for (int c = 0; c < CV; ++c) {
                    const float x = (float)xv[u][c];
#pragma unroll
                    for (int r = 0; r < RP; ++r) acc[c][r] += x * tf[r];
                }
            }
        }
    }
    if (rt > 0 && rt < RT) {
        float* dst = red + ((rt - 1) * CGB + cgl) * (CV * RP + 1);
#pragma unroll
        for (int c = 0; c < CV; ++c)
#pragma unroll
            for (int r = 0; r < RP; ++r) dst[c * RP + r] = acc[c][r];
    }
    __syncthreads();
    if (rt == 0 && n0 < N) {
#pragma unroll
        for (int c = 0; c < CV; ++c)
#pragma unroll
            for (int r = 0; r < RP; ++r) {
                float v = acc[c][r];
#pragma unroll
                for (int k = 0; k < RT - 1; ++k) v += red[(k * CGB + cgl) * (CV * RP + 1) + c * RP + r];
                partial[((int64_t)blockIdx.y * N + n0 + c) * RP + r] = v;
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
    // small ranks: vectorised kernel, 320 columns per block; otherwise 64 columns per block
    const bool vec = (RP == 8 && (N & 7) == 0 && (ldx & 7) == 0) || (RP == 16 && (N & 3) == 0 && (ldx & 3) == 0);
    const int cols_per_block = vec ? (RP == 8 ? 320 : 160) : 64;
    const int ncb = (N + cols_per_block - 1) / cols_per_block;
    int nsplit = (768 + ncb - 1) / ncb;
    if (nsplit > (M + 63) / 64) nsplit = (M + 63) / 64;
    while (nsplit > 1 && (int64_t)nsplit * N * RP > scratch_elems) nsplit >>= 1;
    FD_REQUIRE((int64_t)nsplit * N * RP <= scratch_elems, "fd_lora_wgrad: scratch too small");
    int rows = (M + nsplit - 1) / nsplit;
    rows = vec ? (rows + 5) / 6 * 6 : (rows + 3) & ~3;
    nsplit = (M + rows - 1) / rows;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(ncb, nsplit);
    switch (vec ? -RP : RP) {
        case -8: hipLaunchKernelGGL((lora_wgrad_partial_vec<8, 8, 40>), grid, dim3(256), 0, s, (const f16*)X, ldx, (const f16*)T, ldt, scratch, M, N, rows); break;
        case -16: hipLaunchKernelGGL((lora_wgrad_partial_vec<16, 4, 40>), grid, dim3(256), 0, s, (const f16*)X, ldx, (const f16*)T, ldt, scratch, M, N, rows); break;
        case 8: hipLaunchKernelGGL(lora_wgrad_partial<8>, grid, dim3(256), 0, s, (const f16*)X, ldx, (const f16*)T, ldt, scratch, M, N, rows); break;
        case 16: hipLaunchKernelGGL(lora_wgrad_partial<16>, grid, dim3(256), 0, s, (const f16*)X, ldx, (const f16*)T, ldt, scratch, M, N, rows); break;
        case 32: hipLaunchKernelGGL(lora_wgrad_partial<32>, grid, dim3(256), 0, s, (const f16*)X, ldx, (const f16*)T, ldt, scratch, M, N, rows); break;
        default: hipLaunchKernelGGL(lora_wgrad_partial<64>, grid, dim3(256), 0, s, (const f16*)X, ldx, (const f16*)T, ldt, scratch, M, N, rows); break;
    }
    hipLaunchKernelGGL(lora_wgrad_final, dim3((N * R + 3) / 4), dim3(256), 0, s, scratch, G, g_stride_n, g_stride_r, N, R, RP, nsplit, scale);
    return fd_check_launch("fd_lora_wgrad");
}
