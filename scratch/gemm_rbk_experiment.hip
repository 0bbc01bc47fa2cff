// The K = 320 "weight panel resident in registers" form of the register-B GEMM (gemm_rb.hip, second half): its own translation unit because it wants the
// OTHER register split -- all 200 B registers in AGPRs (constraint "a"), the 120 accumulators in VGPRs (-mllvm -amdgpu-mfma-vgpr-form, csrc/Makefile):
// compiled with the accumulators in AGPRs the allocator spilled the long-lived panel (107 registers of scratch).
#define RBK_ONLY 1
#define RB_AG_FROM_VALUE 0
#include "gemm_rb.hip"
