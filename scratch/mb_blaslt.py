"""Calibration: the library's dense GEMM kernels against torch's (hipBLASLt / rocBLAS) on the step's dense shapes, isolated, warm and cold A.  TFLOP/s."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops
import torch.nn.functional as F
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
shapes = [(65536, 320, 320), (65536, 960, 320), (65536, 2560, 320), (65536, 320, 1280), (16384, 640, 640), (16384, 1920, 640), (16384, 5120, 640), (16384, 640, 2560),
          (4096, 1280, 1280), (4096, 3840, 1280), (4096, 10240, 1280), (4096, 1280, 5120), (1024, 1280, 1280), (1024, 1280, 5120), (65536, 320, 2880), (16384, 640, 5760)]
print("M N K | ours us TF | torch us TF | ours cold-A us | torch cold-A us")
for M, N, K in shapes:
    POOL = max(2, int(600e6 // (M * K * 2)) + 1)
    As = [(torch.randn(M, K, device=dev) * 0.5).half() for _ in range(POOL)]
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half(); bias = torch.randn(N, device=dev)
    b16 = bias.half()
    gf = 2.0 * M * N * K / 1e9
    t0 = timeit(lambda: ops.gemm(As[0], w, bias=bias)); t1 = timeit(lambda: F.linear(As[0], w, b16))
    i = [0]
    def cold(fn):
        def g():
            i[0] = (i[0] + 1) % POOL
            return fn(As[i[0]])
        return g
    c0 = timeit(cold(lambda a: ops.gemm(a, w, bias=bias))); c1 = timeit(cold(lambda a: F.linear(a, w, b16)))
    print(f"{M} {N} {K} | {t0:.1f} {gf / t0 * 1e3:.0f} | {t1:.1f} {gf / t1 * 1e3:.0f} | {c0:.1f} {gf / c0 * 1e3:.0f} | {c1:.1f} {gf / c1 * 1e3:.0f}", flush=True)
    del As
