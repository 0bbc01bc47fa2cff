// Calibration: what does the matrix pipe of one MI355X actually sustain?
//  (1) register-only v_mfma_f32_16x16x32_f16 loop, NACC independent accumulators per wave, W waves per CU
//  (2) the same with the A/B fragments re-read from LDS every k-step (ds_read_b128), i.e. the GEMM main loop minus global loads
// build: hipcc -O3 --offload-arch=gfx950 scratch/mb_mfma_peak.hip -o scratch/mb_mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NA, int NB, bool LDS, int NWAVE, bool BAR>
__global__ __launch_bounds__(NWAVE * 64) void mfma_loop(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 acc[NA][NB];
    for (int i = 0; i < NA; i++) for (int j = 0; j < NB; j++) acc[i][j] = f4{0, 0, 0, 0};
    h8 a[NA], b[NB];
    for (int i = 0; i < NA; i++) for (int e = 0; e < 8; e++) a[i][e] = (_Float16)(0.001f * (lane + i + e));
    for (int j = 0; j < NB; j++) for (int e = 0; e < 8; e++) b[j][e] = (_Float16)(0.002f * (lane + j - e));
    if (LDS) {
        // 64 KB of fragments: wave-private slabs, 16 B per lane, conflict-free
        h8* s = (h8*)smem;
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = a[0];
        __syncthreads();
    }
    for (int it = 0; it < iters; it++) {
        if (BAR && !(it & 1)) __syncthreads();   // one barrier per BK=64 step, as in the GEMM
        if (LDS) {
            const h8* s = (const h8*)smem + ((it & 1) * 2048) + lane;
            #pragma unroll
            for (int i = 0; i < NA; i++) a[i] = s[((wave & 1) * NA + i) * 64];
            #pragma unroll
            for (int j = 0; j < NB; j++) b[j] = s[((16 + (wave >> 1) * NB + j) & 31) * 64];
        }
        #pragma unroll
        for (int i = 0; i < NA; i++)
            #pragma unroll
            for (int j = 0; j < NB; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    f4 s = f4{0, 0, 0, 0};
    for (int i = 0; i < NA; i++) for (int j = 0; j < NB; j++) s += acc[i][j];
    if (s[0] == 12345.678f) out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NA, int NB, bool LDS, int NWAVE = 8, bool BAR = false>
static void run(const char* name, int waves, int blocks) {
    float* out; hipMalloc(&out, 4096);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    size_t sh = LDS ? 65536 : 0;
    hipFuncSetAttribute((const void*)mfma_loop<NA, NB, LDS, NWAVE, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<NA, NB, LDS, NWAVE, BAR>), dim3(blocks), dim3(waves * 64), sh, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = 2.0 * 16 * 16 * 32 * NA * NB * (double)iters * waves * blocks;
        if (rep == 2) printf("%-44s waves/WG %2d blocks %4d  %8.3f ms  %8.1f TFLOP/s\n", name, waves, blocks, ms, flop / ms * 1e-9);
    }
    hipFree(out);
}

// ---- (3) reconciliation with MI355X_MICROARCH.md (2495 TF measured with 32x32x16): shape x data fill x effective clock.
// The chip clocks to its power budget (guide, "DVFS give-back"): zero operands run at ~2.3-2.4 GHz, random operands at ~1.9-2.0 GHz.
typedef float f16v __attribute__((ext_vector_type(16)));
template <int SHAPE>   // 16: v_mfma_f32_16x16x32_f16 (8x5 accumulators)   32: v_mfma_f32_32x32x16_f16 (4x2 accumulators of 16 regs)
__global__ __launch_bounds__(512) void mfma_clock(float* out, long long* cyc, int iters, float fill) {
    const int lane = threadIdx.x & 63;
    h8 a[8], b[5];
    for (int i = 0; i < 8; i++) for (int e = 0; e < 8; e++) a[i][e] = (_Float16)(fill * (0.37f + 0.001f * ((lane * 7 + i * 3 + e * 5) % 97)) * ((lane + e + i) & 1 ? 1.f : -1.f));
    for (int j = 0; j < 5; j++) for (int e = 0; e < 8; e++) b[j][e] = (_Float16)(fill * (0.21f + 0.002f * ((lane * 5 + j * 11 + e * 3) % 89)) * ((lane + e + j) & 2 ? 1.f : -1.f));
    const long long t0 = clock64();
    float keep = 0.f;
    if (SHAPE == 16) {
        f4 acc[8][5];
        for (int i = 0; i < 8; i++) for (int j = 0; j < 5; j++) acc[i][j] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; it++)
            #pragma unroll
            for (int i = 0; i < 8; i++)
                #pragma unroll
                for (int j = 0; j < 5; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        for (int i = 0; i < 8; i++) for (int j = 0; j < 5; j++) keep += acc[i][j][0] + acc[i][j][3];
    } else {
        f16v acc[4][2];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; it++)
            #pragma unroll
            for (int i = 0; i < 4; i++)
                #pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) keep += acc[i][j][0] + acc[i][j][15];
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (keep == 12345.678f) out[threadIdx.x] = keep;
}

template <int SHAPE>
static void run_clock(const char* name, float fill) {
    float* out; long long* cyc; hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    const int iters = SHAPE == 16 ? 20000 : 50000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_clock<SHAPE>), dim3(256 * 4), dim3(512), 0, 0, out, cyc, iters, fill);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double per_wave = SHAPE == 16 ? 2.0 * 16 * 16 * 32 * 40 : 2.0 * 32 * 32 * 16 * 8;
        const double flop = per_wave * (double)iters * 8 * 256 * 4;
        // block 0's wave runs 1/4 of the launch's wall time (4 blocks per CU run back to back): clock = its cycles / (ms / 4)
        if (rep == 2) printf("%-52s %8.3f ms  %8.1f TFLOP/s   one-block cycles %lld (s_memtime ticks; 100 MHz reference clock on this part if << GHz)\n", name, ms, flop / ms * 1e-9, c);
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run_clock<16>("16x16x32 f16, 8x5 acc, ZERO operands", 0.f);
    run_clock<16>("16x16x32 f16, 8x5 acc, signed non-trivial operands", 1.f);
    run_clock<32>("32x32x16 f16, 4x2 acc, ZERO operands", 0.f);
    run_clock<32>("32x32x16 f16, 4x2 acc, signed non-trivial operands", 1.f);
    run<8, 5, false>("regs only 8x5 acc (conv 8-wave shape)", 8, 256);
    run<8, 5, false>("regs only 8x5 acc, 2 WG/CU", 8, 512);
    run<4, 5, false, 16>("regs only 4x5 acc (16-wave shape)", 16, 256);
    run<4, 4, false, 16>("regs only 4x4 acc", 16, 256);
    run<4, 4, false>("regs only 4x4 acc 4 waves", 4, 256);
    run<8, 5, true>("LDS frags 8x5 acc (conv 8-wave shape)", 8, 256);
    run<4, 5, true, 16>("LDS frags 4x5 acc (16-wave shape)", 16, 256);
    run<8, 5, true, 8, true>("LDS frags 8x5 + barrier per 2 k-steps", 8, 256);
    run<4, 5, true, 16, true>("LDS frags 4x5 + barrier per 2 k-steps, 16 waves", 16, 256);
    run<4, 4, true, 16>("LDS frags 4x4 acc", 16, 256);
    // long run to see the sustained (power-capped) clock
    for (int k = 0; k < 3; k++) run<8, 5, false>("regs only 8x5 acc (repeat, sustained)", 8, 256 * 16);
    return 0;
}
