import sys; sys.path.insert(0, "/root/repo")
import torch
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda")
def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M, K in [(65536, 320), (16384, 640), (4096, 1280), (1024, 1280), (65536, 960)]:
    for N in (8, 24):
        x = torch.randn(M, K, device=dev).half()
        w = torch.randn(N, K, device=dev).half()
        us = bench(lambda: ops.gemm(x, w))
        print(f"M={M} K={K} N={N}: {us:.1f} us  ({M*K*2/us/1e6:.2f} TB/s read)")
