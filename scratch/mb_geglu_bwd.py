"""FD_ACT_GEGLU_BWD (the gate's backward in the FF2 data-gradient GEMM's epilogue) against fd_gemm + fd_geglu_bwd_interleaved, isolated, cold operands; us per call."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=20, rep=5):
    for _ in range(3): fn()
    ts = []
    for _ in range(rep):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1000 / n)
    return sorted(ts)[len(ts) // 2]
for M, F, K in [(65536, 1280, 320), (16384, 2560, 640), (4096, 5120, 1280), (1024, 5120, 1280)]:
    POOL = 3
    dh = [(torch.randn(M, K, device=dev) * 0.5).half() for _ in range(POOL)]; pr = [(torch.randn(M, 2 * F, device=dev)).half() for _ in range(POOL)]
    wT = (torch.randn(F, K, device=dev) * K ** -0.5).half()
    i = [0]
    def fused():
        i[0] = (i[0] + 1) % POOL
        return ops.gemm(dh[i[0]], wT, act="geglu_bwd", aux=pr[i[0]])
    def sep():
        i[0] = (i[0] + 1) % POOL
        return ops.geglu_bwd_interleaved(pr[i[0]], ops.gemm(dh[i[0]], wT))
    def g_only():
        i[0] = (i[0] + 1) % POOL
        return ops.gemm(dh[i[0]], wT)
    print(f"{M} x {F} x {K}: fused {timeit(fused):.1f} us   gemm + geglu_bwd {timeit(sep):.1f} us   (gemm alone {timeit(g_only):.1f})", flush=True)
    del dh, pr
