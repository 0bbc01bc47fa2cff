import sys; sys.path.insert(0, "/root/repo")
import torch
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda")
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (M, F, K) in [(65536, 1280, 320), (16384, 2560, 640), (4096, 5120, 1280), (1024, 5120, 1280)]:
    xs = [torch.randn(M, K, device=dev).half() for _ in range(4)]
    w = (torch.randn(2 * F, K, device=dev) * 0.05).half(); bias = torch.randn(2 * F, device=dev)
    wi, bi = ops.interleave_geglu(w, bias)
    i = [0]
    def unf():
        i[0] += 1; return ops.geglu(ops.gemm(xs[i[0] % 4], w, bias=bias))
    def fus():
        i[0] += 1; return ops.gemm(xs[i[0] % 4], wi, bias=bi, act="geglu")
    print(f"FF1+GEGLU M={M} F={F} K={K}: unfused {bench(unf):.1f} us   fused {bench(fus):.1f} us")
