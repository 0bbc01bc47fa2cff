"""gemm_l2_kernel (two 128x320 workgroups per CU, bench-hooks library, FD_GEMM_L2=1) against the shipped dense kernels on the step's short-K shapes; run twice, with and
without the switch (the policy is read once per process)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=30, rep=5):
    for _ in range(3): fn()
    ts = []
    for _ in range(rep):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1000 / n)
    return sorted(ts)[len(ts) // 2]
print("FD_GEMM_L2 =", os.environ.get("FD_GEMM_L2"))
for M, N, K, res in [(65536, 320, 320, False), (65536, 320, 320, True), (65536, 960, 320, False), (65536, 320, 1280, True), (16384, 640, 640, True), (16384, 1920, 640, False), (16384, 640, 2560, True),
                     (4096, 1280, 1280, True), (4096, 3840, 1280, False)]:
    POOL = max(2, int(600e6 // (M * K * 2)) + 1)
    As = [(torch.randn(M, K, device=dev) * 0.5).half() for _ in range(POOL)]
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half(); bias = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev).half() if res else None
    i = [0]
    def f():
        i[0] = (i[0] + 1) % POOL
        return ops.gemm(As[i[0]], w, bias=bias, residual=r)
    t = timeit(f)
    y = ops.gemm(As[0], w, bias=bias, residual=r)
    ref = As[0].float() @ w.float().t() + bias + (r.float() if res else 0)
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    print(f"{M} {N} {K} res={int(res)}: {t:.1f} us {2.0 * M * N * K / t / 1e6:.0f} TF err {err:.1e}", flush=True)
    del As
