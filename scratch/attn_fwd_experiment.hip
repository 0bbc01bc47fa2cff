// Where does the d = 40 attention forward spend its time?  Ablations of the shipped kernel (copied below with MODE switches; results are
// garbage for MODE != 0, only the time matters) on the step's shape: B = 16, H = 8, T = 4096, d = 40, q/k as column slices of a [M, 3C]
// buffer.  build: hipcc -O3 --offload-arch=gfx950 scratch/attn_fwd_experiment.hip -o scratch/attn_fwd_experiment
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../finetune_fair_diffusion_amd/csrc/attn.hip"
#include "attn_fwd_glds_experiment.h"
void fd_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int fd_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -1; }

#ifndef GNW
#define GNW 4
#endif
enum { NO_EXP = 1, NO_PV = 2, NO_QK = 4, NO_GLOAD = 8, NO_STAGE = 16, NO_VREAD = 32, NO_MAX = 64 };

template <int D, int MODE>
__global__ __launch_bounds__(256, 3) void attn_fwd_abl(const f16* __restrict__ Q, const f16* __restrict__ K, const f16* __restrict__ Vt,
                                                    f16* __restrict__ O, float* __restrict__ LSE, int H, int Tq, int Tk, int Tkp,
                                                    int Tkr, int kv_div, float scale, int ldq, int ldk) {
    constexpr int DK = (D + 15) / 16 * 16, DV = (D + 31) / 32 * 32, DKP = DK + 8;
    constexpr int NKS = DK / 16, NDV = DV / 32;
    extern __shared__ __attribute__((aligned(16))) f16 smem[];
    f16* Ks = smem;
    f16* Vts = smem + 64 * DKP;
    int b, h, qblk;
    attn_block_coords((Tq + 127) / 128, H, gridDim.x / (((Tq + 127) / 128) * H), b, h, qblk);
    const int q0 = qblk * 128;
    const int bk = b / kv_div;
    const int C = H * D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 31, g = lane >> 5;
    const int t = q0 + wave * 32 + ql;
    const bool tvalid = t < Tq;
    f16x8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int col = ks * 16 + g * 8;
        qf[ks] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (tvalid && col < D) qf[ks] = *(const f16x8*)(Q + ((int64_t)b * Tq + t) * ldq + h * D + col);
    }
    f32x16 oacc[NDV];
#pragma unroll
    for (int i = 0; i < NDV; ++i) oacc[i] = zero16();
    float m_run = -INFINITY, l_run = 0.f;
    const float sl2 = scale * LOG2E;
    const f16* Kb = K + (int64_t)bk * Tkr * ldk + h * D;
    const f16* Vtb = Vt + ((int64_t)bk * C + h * D) * Tkp;
    TileRegs<D> kreg, vreg;
    TilePlan<D> kplan, vplan;
    plan_rows<D>(kplan, ldk);
    plan_cols<D>(vplan, Tkp);
    zero_row_pad<D, DKP>(Ks);
    zero_col_pad<D, DV>(Vts);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    load_rows<D>(kreg, Kb, ldk, 0, Tk);
    load_cols<D>(vreg, Vtb, Tkp, 0, Tkp);
    if (MODE & NO_STAGE) {
        __syncthreads();
        store_rows<D, DKP>(kreg, Ks);
        store_cols<D>(vreg, Vts);
        __syncthreads();
    }
    for (int k0 = 0; k0 < Tk; k0 += 64) {
        if (!(MODE & NO_STAGE)) {
            __syncthreads();
            store_rows<D, DKP>(kreg, Ks);
            store_cols<D>(vreg, Vts);
            __syncthreads();
        }
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            s[kt] = zero16();
            if (MODE & NO_QK) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kt][r] = (float)qf[0][r & 7] + (float)(k0 + r);
            } else {
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const f16x8 kf = *(const f16x8*)(Ks + (kt * 32 + ql) * DKP + ks * 16 + g * 8);
                    s[kt] = mfma32(kf, qf[ks], s[kt]);
                }
            }
        }
        if (!(MODE & NO_GLOAD)) {
            if (k0 + 128 <= Tk) {
                load_planned<D>(kreg, Kb + (int64_t)(k0 + 64) * ldk, kplan);
                load_planned<D>(vreg, Vtb + (k0 + 64), vplan);
            } else if (k0 + 64 < Tk) {
                load_rows<D>(kreg, Kb, ldk, k0 + 64, Tk);
                load_cols<D>(vreg, Vtb, Tkp, k0 + 64, Tkp);
            }
        }
        float m_new, alpha;
        if (MODE & NO_MAX) {
            m_new = 4.f; alpha = 1.f;
        } else {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            m_new = fmaxf(m_run, mx);
            alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2);
        }
        const float nm = -m_new * sl2;
        float rs = 0.f;
        f16x8 pf[4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a = fmaf(s[kt][r], sl2, nm);
                const float p = (MODE & NO_EXP) ? a : __builtin_amdgcn_exp2f(a);
                rs += p;
                pf[kt * 2 + (r >> 3)][r & 7] = (f16)p;
            }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        if (!(MODE & NO_MAX) && __any(m_new != m_run)) {
#pragma unroll
            for (int i = 0; i < NDV; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        }
        m_run = m_new;
        if (MODE & NO_PV) {
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int j = 0; j < 8; ++j) oacc[0][(st * 8 + j) & 15] += (float)pf[st][j];
        } else {
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int i = 0; i < NDV; ++i) {
                    const f16x8 vf = (MODE & NO_VREAD) ? qf[(st + i) % NKS] : read_perm(Vts, i * 32 + ql, st * 16, g);
                    oacc[i] = mfma32(vf, pf[st], oacc[i]);
                }
        }
    }
    if (tvalid) {
        const float inv = 1.f / l_run;
        f16* Op = O + ((int64_t)b * Tq + t) * C + h * D;
#pragma unroll
        for (int i = 0; i < NDV; ++i)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int dv = i * 32 + 8 * rq + 4 * g;
                if (dv < D) {
                    f16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (f16)(oacc[i][rq * 4 + j] * inv);
                    *(f16x4*)(Op + dv) = o;
                }
            }
        if (LSE && g == 0) LSE[((int64_t)b * H + h) * Tq + t] = m_run * scale + log2f(l_run) / LOG2E;
    }
}

static f16 *q, *vt, *o;
static float* lse;
static const int B = 16, H = 8, T = 4096, D = 40, C = H * D;

template <int MODE>
static void run(const char* name) {
    dim3 grid(((T + 127) / 128) * H * B);
    auto launch = [&]() {
        hipLaunchKernelGGL((attn_fwd_abl<D, MODE>), grid, dim3(256), fwd_lds<D>(), 0, q, q + C, vt, o, lse, H, T, T, T, T, 1, 0.158f, 3 * C, 3 * C);
    };
    for (int i = 0; i < 3; ++i) launch();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int n = 20;
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / n, fl = 4.0 * B * H * (double)T * T * D;
    printf("%-44s %8.1f us  %7.1f TFLOP/s (useful)\n", name, us, fl / us * 1e-6);
}

int main() {
    const size_t nq = (size_t)B * T * 3 * C;
    std::vector<f16> hq(nq);
    unsigned s = 12345;
    for (auto& x : hq) { s = s * 1664525u + 1013904223u; x = (f16)(((int)(s >> 16) % 2001 - 1000) * 0.002f); }
    hipMalloc(&q, nq * 2); hipMalloc(&vt, (size_t)B * C * T * 2); hipMalloc(&o, (size_t)B * T * C * 2); hipMalloc(&lse, (size_t)B * H * T * 4);
    hipMemcpy(q, hq.data(), nq * 2, hipMemcpyHostToDevice);
    hipMemcpy(vt, hq.data(), (size_t)B * C * T * 2, hipMemcpyHostToDevice);
    {   // the shipped kernel through its C entry point, for reference
        for (int i = 0; i < 3; ++i) fd_attn_fwd(q, q + C, vt, o, lse, B, H, T, T, T, T, D, 1, 0.158f, 3 * C, 3 * C, nullptr);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) fd_attn_fwd(q, q + C, vt, o, lse, B, H, T, T, T, T, D, 1, 0.158f, 3 * C, 3 * C, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f us\n", "shipped fd_attn_fwd", ms * 1e3 / 20);
    }
    {   // the backward kernels on the same shape (q/k/v/dq/dk/dv as slices of [M, 3C] buffers)
        f16 *dqkv, *dot_, *qt;
        float* Dd;
        hipMalloc(&dqkv, nq * 2); hipMalloc(&dot_, (size_t)B * C * T * 2); hipMalloc(&qt, (size_t)B * C * T * 2); hipMalloc(&Dd, (size_t)B * H * T * 4);
        hipMemcpy(dot_, hq.data() + 777, (size_t)B * C * T * 2, hipMemcpyHostToDevice);
        hipMemcpy(qt, hq.data() + 1555, (size_t)B * C * T * 2, hipMemcpyHostToDevice);
        fd_attn_bwd_prep(o, o, Dd, B, H, T, D, nullptr);
        auto timeit = [&](const char* name, auto fn) {
            for (int i = 0; i < 3; ++i) fn();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) fn();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-44s %8.1f us\n", name, ms * 1e3 / 10);
        };
        timeit("shipped fd_attn_bwd_dq", [&]() { fd_attn_bwd_dq(q, q + C, q + 2 * C, vt, o, lse, Dd, o, dqkv, B, H, T, T, T, T, D, 1, 0.158f, 3 * C, 3 * C, 3 * C, nullptr); });
        timeit("shipped fd_attn_bwd_dkdv", [&]() { fd_attn_bwd_dkdv(q, qt, q + C, q + 2 * C, o, dot_, lse, Dd, dqkv + C, dqkv + 2 * C, B, H, T, T, T, D, 1, 0.158f, 3 * C, 3 * C, 3 * C, nullptr); });
        timeit("shipped fd_attn_fwd (again)", [&]() { fd_attn_fwd(q, q + C, vt, o, lse, B, H, T, T, T, T, D, 1, 0.158f, 3 * C, 3 * C, nullptr); });
    }
    {   // register-staged kernel vs the direct-to-LDS pipeline: same inputs, compare O and LSE, time both
        f16* o2; float* lse2;
        hipMalloc(&o2, (size_t)B * T * C * 2); hipMalloc(&lse2, (size_t)B * H * T * 4);
        dim3 grid(((T + 127) / 128) * H * B);
        hipFuncSetAttribute((const void*)attn_fwd_glds_kernel<D, GNW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_glds_lds<D>());
        dim3 grid2(((T + 32 * GNW - 1) / (32 * GNW)) * H * B);
        auto old_k = [&]() { hipLaunchKernelGGL((attn_fwd_kernel<D>), grid, dim3(256), fwd_lds<D>(), 0, q, q + C, vt, o, lse, H, T, T, T, T, 1, 0.158f, 3 * C, 3 * C); };
        auto new_k = [&]() { hipLaunchKernelGGL((attn_fwd_glds_kernel<D, GNW>), grid2, dim3(64 * GNW), fwd_glds_lds<D>(), 0, q, q + C, vt, o2, lse2, H, T, T, T, T, 1, 0.158f, 3 * C, 3 * C); };
        old_k(); new_k(); hipDeviceSynchronize();
        printf("launch status: %s\n", hipGetErrorString(hipGetLastError()));
        std::vector<f16> ho((size_t)B * T * C), ho2((size_t)B * T * C);
        std::vector<float> hl((size_t)B * H * T), hl2((size_t)B * H * T);
        hipMemcpy(ho.data(), o, ho.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(ho2.data(), o2, ho2.size() * 2, hipMemcpyDeviceToHost);
        hipMemcpy(hl.data(), lse, hl.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hl2.data(), lse2, hl2.size() * 4, hipMemcpyDeviceToHost);
        double mo = 0, ml = 0, amax = 0; size_t nbad = 0;
        for (size_t i = 0; i < ho.size(); ++i) { double a = (float)ho[i], b2 = (float)ho2[i]; if (!(b2 == b2)) ++nbad; mo = fmax(mo, fabs(a - b2)); amax = fmax(amax, fabs(a)); }
        for (size_t i = 0; i < hl.size(); ++i) ml = fmax(ml, fabs((double)hl[i] - hl2[i]));
        printf("glds vs staged: max|dO| = %.3e (max|O| = %.3f, NaNs %zu), max|dLSE| = %.3e\n", mo, amax, nbad, ml);
        for (int rep = 0; rep < 2; ++rep) {
            for (auto kn : {0, 1}) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                for (int i = 0; i < 3; ++i) { if (kn) new_k(); else old_k(); }
                hipEventRecord(e0);
                for (int i = 0; i < 20; ++i) { if (kn) new_k(); else old_k(); }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double us = ms * 1e3 / 20, fl = 4.0 * B * H * (double)T * T * D;
                printf("%-44s %8.1f us  %7.1f TFLOP/s (useful)\n", kn ? "attn_fwd_glds_kernel (3-buffer direct-to-LDS)" : "attn_fwd_kernel (register-staged)", us, fl / us * 1e-6);
            }
        }
    }
    if (getenv("ABL") == nullptr) return 0;
    run<0>("full");
    run<NO_EXP>("no v_exp");
    run<NO_MAX>("no running max / rescale");
    run<NO_EXP | NO_MAX>("no v_exp, no max");
    run<NO_PV>("no PV MFMAs (nor V reads)");
    run<NO_VREAD>("PV MFMAs without the V LDS reads");
    run<NO_QK>("no QK MFMAs (nor K reads)");
    run<NO_GLOAD>("no global loads in the loop");
    run<NO_STAGE>("no LDS staging / barriers in the loop");
    run<NO_GLOAD | NO_STAGE>("no loads, no staging");
    run<NO_QK | NO_PV>("softmax only (no MFMA)");
    run<NO_EXP | NO_MAX | NO_GLOAD | NO_STAGE>("MFMAs + LDS fragment reads + cvt only");
    return 0;
}
