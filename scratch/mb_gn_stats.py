"""Isolated cost of the GroupNorm-statistics epilogue: the same launch with / without fd_gemm_desc.gn_stats (HIP events, hot caches)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda", 0)
def t(fn, n=40):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n
g = torch.Generator().manual_seed(0)
for (B, H, Cin, Cout) in [(16, 64, 320, 320), (16, 64, 640, 320), (16, 32, 640, 640), (16, 16, 1280, 1280)]:
    x = (torch.randn(B * H * H, Cin, generator=g)).to(dev).half(); w = (torch.randn(Cout, 9 * Cin, generator=g) * 0.05).to(dev).half()
    bias = torch.randn(Cout, generator=g).to(dev)
    for st in (False, True, False, True):
        print(f"conv {Cin}->{Cout} @{H}^2 b{B} stats={st}: {t(lambda: ops.conv3x3(x, w, B, H, H, bias=bias, gn_stats=st)):.1f} us")
for (M, N, K) in [(65536, 320, 320), (32768, 320, 320), (16384, 640, 640)]:
    a = torch.randn(M, K, generator=g).to(dev).half(); b = (torch.randn(N, K, generator=g) * 0.1).to(dev).half(); r = torch.randn(M, N, generator=g).to(dev).half()
    bias = torch.randn(N, generator=g).to(dev)
    for st in (False, True, False, True):
        print(f"gemm {M}x{N}x{K} +res stats={st}: {t(lambda: ops.gemm(a, b, bias=bias, residual=r, gn_stats=st)):.1f} us")
C = 320
for (B, HW) in [(16, 4096)]:
    x = torch.randn(B * HW, C, generator=g).to(dev).half(); gm = torch.ones(C, device=dev); bt = torch.zeros(C, device=dev)
    a = torch.randn(B * HW, C, generator=g).to(dev).half(); wb = (torch.randn(C, C, generator=g) * 0.1).to(dev).half()
    xs = ops.gemm(a, wb, gn_stats=True)
    print(f"groupnorm 16x4096x320 two-launch: {t(lambda: ops.groupnorm(x, None, B, HW, 32, 1e-5, gm, bt, True)):.1f} us   from statistics: {t(lambda: ops.groupnorm(xs, None, B, HW, 32, 1e-5, gm, bt, True)):.1f} us")
