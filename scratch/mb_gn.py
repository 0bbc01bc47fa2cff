import sys; sys.path.insert(0, "/root/repo")
import torch
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda")
def bench(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B = 16
for (H, C) in [(64, 320), (64, 640), (64, 960), (32, 640), (32, 1280), (16, 1280), (16, 2560), (8, 1280)]:
    xs = [torch.randn(B * H * H, C, device=dev).half() for _ in range(6)]
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    i = [0]
    def f():
        i[0] += 1
        return ops.groupnorm(xs[i[0] % 6], None, B, H * H, 32, 1e-5, gamma, beta, True)
    us = bench(f)
    mb = B * H * H * C * 2 / 1e6
    y, st = f()
    dy = torch.randn_like(y)
    def g():
        i[0] += 1
        return ops.groupnorm_bwd(xs[i[0] % 6], None, dy, B, H * H, 32, st, gamma, beta, True)
    us_b = bench(g)
    print(f"GN fwd {H}x{H}x{C}: {us:.1f} us ({3*mb/us/1e6*1e6/1e6:.2f} TB/s over 3x{mb:.0f} MB)   bwd: {us_b:.1f} us ({5*mb/us_b:.2f} TB/s over 5x)")
