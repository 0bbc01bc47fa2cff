"""A/B of the ping-pong 256x320 kernel (csrc/gemm_pp.hip) against the shipped lockstep kernels on the step's shapes, with a correctness check.
Needs the bench-hooks library (make BENCH_HOOKS=1; FAIRDIFF_LIB=.../libfairdiff_hip_bench.so): FD_GEMM_PP is re-read on every call there.
    FD_GEMM_PP bits: 1 conv 256x320, 2 dense 256x320, 4 s_setprio, 8 conv 128x320 (split-K included), 16 dense 128x320 (gemm.hip pp_takes)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dev).half()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


def run(name, fn, flops):
    outs, ts = {}, {}
    for mode in (0, 27, 31):
        os.environ["FD_GEMM_PP"] = str(mode)
        outs[mode] = fn().float()
        ts[mode] = timeit(fn)
    ref = outs[0]
    e3 = float((outs[27] - ref).abs().max() / ref.abs().max())
    e7 = float((outs[31] - ref).abs().max() / ref.abs().max())
    print(f"{name:44s} shipped {ts[0]:7.1f} us ({flops / ts[0] / 1e6:6.0f} TF/s)  pp {ts[27]:7.1f} us ({flops / ts[27] / 1e6:6.0f})  pp+prio {ts[31]:7.1f} us ({flops / ts[31] / 1e6:6.0f})"
          f"   max rel diff vs shipped {e3:.1e} / {e7:.1e}", flush=True)
    assert e3 < 2e-3 and e7 < 2e-3, name


def conv(B, H, Cin, Cout, residual=False):
    x, w = rnd(B * H * H, Cin), rnd(Cout, 9 * Cin, scale=(9 * Cin) ** -0.5)
    bias = torch.randn(Cout, generator=g).to(dev)
    res = rnd(B * H * H, Cout) if residual else None
    run(f"conv {Cin}->{Cout} @{H}^2 B{B}{' +res' if residual else ''}", lambda: ops.conv3x3(x, w, B, H, H, bias=bias, residual=res)[0], 2.0 * B * H * H * Cout * 9 * Cin)


def gemm(M, N, K, residual=False, k2=0, act="none"):
    a, b = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    bias = torch.randn(N, generator=g).to(dev)
    res = rnd(M, N) if residual else None
    a2, b2 = (rnd(M, k2), rnd(N, k2, scale=0.1)) if k2 else (None, None)
    run(f"gemm {M}x{N}x{K}{'+' + str(k2) if k2 else ''}{' +res' if residual else ''} {act if act != 'none' else ''}",
        lambda: ops.gemm(a, b, a2=a2, b2=b2, bias=bias, residual=res, act=act), 2.0 * M * N * (K + k2))


if __name__ == "__main__":
    # correctness first, on awkward sizes: M not a multiple of 256, K not a multiple of 64, image borders, LoRA slab
    conv(1, 24, 320, 320)
    conv(3, 16, 352, 320, residual=True)
    gemm(4096 + 40, 320, 352)
    gemm(25600, 640, 320, residual=True, k2=8)
    conv(2, 16, 1280, 1280)                 # 128x320 tile
    conv(3, 8, 1280, 1280, residual=True)   # 128x320 tile, split-K
    gemm(4096 + 24, 1280, 1280 + 32, residual=True)
    # the step's shapes (CFG batch 16)
    conv(16, 64, 320, 320)
    conv(16, 64, 320, 320, residual=True)
    conv(16, 64, 640, 320)
    conv(16, 64, 960, 320)
    conv(16, 32, 640, 640)
    conv(16, 32, 1280, 640)
    conv(16, 32, 1920, 640)
    gemm(65536, 2560, 320)
    gemm(65536, 2560, 320, act="geglu")
    gemm(65536, 320, 1280, residual=True)
    gemm(65536, 320, 320, residual=True)
    gemm(65536, 960, 320, k2=24)
    gemm(16384, 640, 2560, residual=True)
    gemm(16384, 5120, 640, act="geglu")
    gemm(16384, 640, 640, residual=True)
    gemm(16384, 1920, 640, k2=24)
    # 16^2 / 8^2 levels: the 128x320 tile (8^2: split-K)
    conv(16, 16, 1280, 1280)
    conv(16, 16, 1280, 1280, residual=True)
    conv(16, 16, 2560, 1280)
    conv(16, 16, 1920, 1280)
    conv(16, 8, 1280, 1280)
    conv(16, 8, 2560, 1280)
    gemm(4096, 1280, 5120, residual=True)
    gemm(4096, 1280, 1280, residual=True)
    gemm(4096, 10240, 1280, act="geglu")
    gemm(4096, 3840, 1280, k2=24)
