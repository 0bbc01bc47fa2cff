// Error plumbing shared by every entry point of libfairdiff_hip.so.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void fd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fd_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fd_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return FD_ERR_LAUNCH;
    }
    return FD_OK;
}

extern "C" const char* fd_last_error(void) { return g_err; }
extern "C" int fd_version(void) { return FD_ABI_VERSION; }
extern "C" const char* fd_working_dtype(void) { return FD_WD_NAME; }
// How this library was built, for lib.load(): packed-fp32 VALU instructions return wrong lanes on gfx950 when kernels of several streams share a SIMD
// ("Round 4 at a glance"), so the Makefile passes -DFD_NO_PACKED_FP32 together with the two flags that keep hipcc from emitting them.
#ifdef FD_NO_PACKED_FP32
extern "C" const char* fd_build_info(void) { return "packed_fp32=off"; }
#else
extern "C" const char* fd_build_info(void) { return "packed_fp32=on"; }
#endif

#ifdef FD_BENCH_HOOKS
// measurement builds: install (or remove: buf = nullptr) the workgroup-trace log in every translation unit that has instrumented kernels (common.h)
extern "C" {
void fd_wgt_set_gemm(unsigned long long*, unsigned int); void fd_wgt_set_gemm_pp(unsigned long long*, unsigned int); void fd_wgt_set_gemm_halo(unsigned long long*, unsigned int);
void fd_wgt_set_attn(unsigned long long*, unsigned int); void fd_wgt_set_crossattn(unsigned long long*, unsigned int); void fd_wgt_set_norm(unsigned long long*, unsigned int);
void fd_wgt_set_elementwise(unsigned long long*, unsigned int); void fd_wgt_set_lora(unsigned long long*, unsigned int);
void fd_bench_wg_trace(unsigned long long* buf, unsigned int cap) {
    fd_wgt_set_gemm(buf, cap); fd_wgt_set_gemm_pp(buf, cap); fd_wgt_set_gemm_halo(buf, cap); fd_wgt_set_attn(buf, cap); fd_wgt_set_crossattn(buf, cap);
    fd_wgt_set_norm(buf, cap); fd_wgt_set_elementwise(buf, cap); fd_wgt_set_lora(buf, cap);
}
}
#endif
