import sys; sys.path.insert(0, "/root/repo")
import torch
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda")
def bench(name, fn, flops, n=40):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print(f"{name:40s} {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TFLOP/s")
# shapes the 128x160 / 256x128 tiles take in the step (16x16 level of the U-Net at CFG batch 16 and R3 micro-batches)
for (M, N, K) in [(4096, 1280, 1280), (4096, 1280, 5120), (4096, 1280, 2560), (6144, 640, 640), (4096, 1280, 640), (65536, 128, 1152), (65536, 320, 1280)]:
    a_ = torch.randn(M, K, device=dev).half(); b_ = (torch.randn(N, K, device=dev) * 0.02).half()
    bench(f"gemm {M}x{N}x{K}", lambda: ops.gemm(a_, b_), 2.0 * M * N * K)
