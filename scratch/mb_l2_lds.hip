// Calibration: how fast can one CU pull GEMM operand tiles out of L2, by which path?
//   MODE 0  global_load_lds_dwordx4 (direct to LDS), s_waitcnt vmcnt(0) + barrier per k-tile   (what gemm_big_kernel does)
//   MODE 1  global_load_lds_dwordx4, wait per k-tile, no barrier
//   MODE 2  global_load_dwordx4 -> VGPR -> ds_write_b128, barrier per k-tile                    (classic register staging)
//   MODE 3  global_load_dwordx4 -> VGPR only, wait per k-tile
//   MODE 4  global_load_lds_dwordx4, keeps one k-tile in flight while waiting for the previous (vmcnt(N))
// Tile = 256 A rows + 320 B rows, 128 B per row per k-tile (73.7 KB), like the 256x320x64 fp16 tile.
// build: hipcc -O3 --offload-arch=gfx950 scratch/mb_l2_lds.hip -o scratch/mb_l2_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const char* src, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

template <int NW, int MODE>
__global__ __launch_bounds__(NW * 64) void pull(const char* A, const char* B, int64_t ld, int nk, int reps, int adiv, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = 576, NG = ROWS / 8, PER = (NG + NW - 1) / NW;   // 72 eight-row groups
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lrow = lane >> 3, chunk = (lane & 7) * 16;
    const int wg = blockIdx.x;
    const char* src[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int g = wave + i * NW;
        const int row = g * 8 + lrow;
        src[i] = row < 256 ? A + ((int64_t)(wg % adiv) * 256 + row) * ld + chunk : B + (int64_t)(row - 256) * ld + chunk;
    }
    u4 keep = u4{0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (MODE == 0 || MODE == 1 || MODE == 4) {
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int g = wave + i * NW;
                    if (g < NG) glds16(src[i] + kt * 128, smem + (buf * ROWS + g * 8) * 128);
                }
                if (MODE == 4) {
                    if (PER == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (MODE == 0) __syncthreads();
            } else {
                u4 v[PER];
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int g = wave + i * NW;
                    if (g < NG) v[i] = __builtin_nontemporal_load((const u4*)(src[i] + kt * 128));
                }
                if (MODE == 2) {
#pragma unroll
                    for (int i = 0; i < PER; ++i) {
                        const int g = wave + i * NW;
                        if (g < NG) *(u4*)(smem + (buf * ROWS + g * 8) * 128 + lane * 16) = v[i];
                    }
                    __syncthreads();
                } else {
#pragma unroll
                    for (int i = 0; i < PER; ++i) {
                        const int g = wave + i * NW;
                        if (g < NG) keep ^= v[i];
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (keep[0] == 0x12345678u) out[0] = 1.f;
    if (MODE != 3 && smem[threadIdx.x] == 77 && out[1] == 3.f) out[2] = 1.f;
}

template <int NW, int MODE>
static void run(const char* name, const char* A, const char* B, int64_t ld, int nk, int adiv, int blocks, float* out) {
    const int reps = 40;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t sh = 2 * 576 * 128;
    (void)hipFuncSetAttribute((const void*)pull<NW, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((pull<NW, MODE>), dim3(blocks), dim3(NW * 64), sh, 0, A, B, ld, nk, reps, adiv, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double bytes = 576.0 * 128 * nk * reps * blocks;
    printf("%-58s NW %2d ld %5ld adiv %4d WGs %4d  %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU@2.1GHz\n", name, NW, (long)ld, adiv, blocks, ms, bytes / ms * 1e-9,
           bytes / ms * 1e-9 * 1e12 / 256 / 2.1e9 / (blocks > 256 ? 1 : 256.0 / blocks) * (blocks > 256 ? 1 : 256.0 / blocks));
}

int main() {
    char *A, *B; float* out;
    const size_t abytes = (size_t)256 * 256 * 2560;   // 256 distinct A tiles x K=1280
    (void)hipMalloc(&A, abytes); (void)hipMalloc(&B, 320 * 2560); (void)hipMalloc(&out, 64);
    (void)hipMemset(A, 1, abytes); (void)hipMemset(B, 2, 320 * 2560); (void)hipMemset(out, 0, 64);
    printf("-- L2-resident (8 distinct A tiles, one per XCD)\n");
    run<8, 0>("glds x4, wait+barrier per tile (GEMM today)", A, B, 2560, 20, 8, 256, out);
    run<8, 1>("glds x4, wait per tile, no barrier", A, B, 2560, 20, 8, 256, out);
    run<8, 4>("glds x4, one tile kept in flight", A, B, 2560, 20, 8, 256, out);
    run<8, 2>("global_load x4 -> VGPR -> ds_write, barrier per tile", A, B, 2560, 20, 8, 256, out);
    run<8, 3>("global_load x4 -> VGPR only", A, B, 2560, 20, 8, 256, out);
    run<16, 0>("glds x4, wait+barrier per tile, 16 waves", A, B, 2560, 20, 8, 256, out);
    run<16, 4>("glds x4, one tile in flight, 16 waves", A, B, 2560, 20, 8, 256, out);
    run<16, 2>("global_load -> VGPR -> ds_write, 16 waves", A, B, 2560, 20, 8, 256, out);
    run<16, 3>("global_load -> VGPR only, 16 waves", A, B, 2560, 20, 8, 256, out);
    printf("-- A streamed (256 distinct A tiles = 168 MB per pass: Infinity Cache / HBM), B L2-resident\n");
    run<8, 0>("glds x4, wait+barrier per tile (GEMM today)", A, B, 2560, 20, 256, 256, out);
    run<8, 4>("glds x4, one tile kept in flight", A, B, 2560, 20, 256, 256, out);
    run<8, 3>("global_load x4 -> VGPR only", A, B, 2560, 20, 256, 256, out);
    run<16, 3>("global_load -> VGPR only, 16 waves", A, B, 2560, 20, 256, 256, out);
    return 0;
}
