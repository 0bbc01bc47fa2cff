import sys; sys.path.insert(0, "/root/repo")
import torch
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda")
def bench(name, fn, flops, n=30):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print(f"{name:40s} {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TFLOP/s")
B = 16
for (H, Cin, Cout) in [(64, 320, 320), (32, 640, 640)]:
    x = torch.randn(B * H * H, Cin, device=dev).half()
    w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).half()
    bias = torch.randn(Cout, device=dev)
    bench(f"conv3x3 {Cin}->{Cout}@{H}", lambda: ops.conv3x3(x, w, B, H, H, bias=bias), 2.0 * B * H * H * Cout * 9 * Cin)
for (M, N, K) in [(65536, 320, 1280), (16384, 5120, 640)]:
    a_ = torch.randn(M, K, device=dev).half(); b_ = (torch.randn(N, K, device=dev) * 0.02).half()
    bench(f"gemm {M}x{N}x{K}", lambda: ops.gemm(a_, b_), 2.0 * M * N * K)
