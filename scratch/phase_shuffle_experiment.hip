// The second pass of the two-pass Upsample2D form of rounds 2-3 (FD_CONV_UP2P writes [4][B,H,W,C] phase-major, this kernel interleaves the phases)
// as it stood when it left the product in round 5.  Round 4's FD_CONV_UP2PI writes the interleaved result from the GEMM epilogue,
// bit-identical (asserted by test_conv_up2_phase_decomposition until this removal) and one launch shorter.

// phase-major [4][B,H,W,C] (phase = py*2+px) -> [B,2H,2W,C]: output pixel (2y+py, 2x+px) of the phase-decomposed upsample conv
__global__ void phase_shuffle_kernel(const f16* __restrict__ src, f16* __restrict__ dst, int B, int H, int W, int C) {
    const int CV = C / 8;
    const int64_t n = (int64_t)B * 4 * H * W * CV;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * 8;
        int64_t q = i / CV;                                   // destination pixel index
        const int X = (int)(q % (2 * W)); q /= 2 * W;
        const int Y = (int)(q % (2 * H));
        const int b = (int)(q / (2 * H));
        const int ph = (Y & 1) * 2 + (X & 1);
        const f16x8 v = *(const f16x8*)(src + ((((int64_t)ph * B + b) * H + (Y >> 1)) * W + (X >> 1)) * C + c);
        *(f16x8*)(dst + i * 8) = v;
    }
}
extern "C" int fd_phase_shuffle(const void* src, void* dst, int B, int H, int W, int C, void* stream) {
    FD_REQUIRE((C & 7) == 0, "fd_phase_shuffle: C%%8");
    hipLaunchKernelGGL(phase_shuffle_kernel, grid_for((int64_t)B * 4 * H * W * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const f16*)src, (f16*)dst, B, H, W, C);
    return fd_check_launch("fd_phase_shuffle");
}
