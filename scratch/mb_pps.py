"""A/B of the persistent streaming 128x320 GEMM (csrc/gemm_pps.hip, policy bit 128) against the shipped kernels on the step's dense shapes,
with a bit-exactness check (same k-order and rounding as the other 16x16x32 kernels: results must be EQUAL).
Needs the bench-hooks library (make BENCH_HOOKS=1; FAIRDIFF_LIB=.../libfairdiff_hip_bench.so): FD_GEMM_PP is re-read on every call there;
FD_GEMM_PPS_MIN / FD_GEMM_PPS_WG are read once per process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
BASE = 1 | 4 | 8 | 32


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dev).half()


def timeit(fn, n=30):
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


def gemm(M, N, K, residual=False, k2=0, act="none", bias=True):
    a, b = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    bv = torch.randn(N, generator=g).to(dev) if bias else None
    n_out = N // 2 if act == "geglu" else N
    res = rnd(M, n_out) if residual else None
    a2, b2 = (rnd(M, k2), rnd(N, k2, scale=0.1)) if k2 else (None, None)
    fn = lambda: ops.gemm(a, b, a2=a2, b2=b2, bias=bv, residual=res, act=act)
    outs, ts = {}, {}
    for mode in (BASE, BASE | 128):
        os.environ["FD_GEMM_PP"] = str(mode)
        outs[mode] = fn().float()
        torch.cuda.synchronize()
        ts[mode] = timeit(fn)
    ref, got = outs[BASE], outs[BASE | 128]
    bad = int((ref != got).sum())
    fl = 2.0 * M * N * (K + k2)
    name = f"gemm {M}x{N}x{K}{'+' + str(k2) if k2 else ''}{' +res' if residual else ''}{' ' + act if act != 'none' else ''}{'' if bias else ' nobias'}"
    print(f"{name:44s} shipped {ts[BASE]:7.1f} us ({fl / ts[BASE] / 1e6:6.0f} TF/s)   streaming {ts[BASE | 128]:7.1f} us ({fl / ts[BASE | 128] / 1e6:6.0f})"
          f"   {100 * (ts[BASE] / ts[BASE | 128] - 1):+5.1f} %   differing outputs {bad}", flush=True)
    assert bad == 0 and torch.isfinite(got).all(), name


if __name__ == "__main__":
    # awkward sizes first: ragged M, K not a multiple of 32 / 64, LoRA slab, several column blocks per workgroup range, fewer tiles than workgroups
    gemm(128 * 300 + 40, 320, 352)
    gemm(128 * 257, 640, 320, residual=True, k2=8)
    gemm(128 * 200 + 8, 960, 320, k2=24, bias=False)
    gemm(128 * 513, 320, 320, residual=True)
    gemm(128 * 100, 2560, 320, act="geglu")
    gemm(128 * 64 + 100, 1280, 640, residual=True)
    # the step's dense shapes (CFG batch 16 and per-rollout batch 8)
    for M in (65536, 32768):
        gemm(M, 320, 320)
        gemm(M, 320, 320, residual=True)
        gemm(M, 960, 320, k2=24, bias=False)
        gemm(M, 320, 320, k2=8)
        gemm(M, 320, 1280, residual=True)
        gemm(M, 2560, 320)
        gemm(M, 2560, 320, act="geglu")
        gemm(M, 320, 768 if False else 320, residual=True, k2=8)
    for M in (16384, 8192):
        gemm(M, 640, 640)
        gemm(M, 640, 640, residual=True)
        gemm(M, 1920, 640, k2=24, bias=False)
        gemm(M, 640, 2560, residual=True)
        gemm(M, 5120, 640)
        gemm(M, 5120, 640, act="geglu")
    gemm(4096, 10240, 1280, act="geglu")
    gemm(4096, 3840, 1280, k2=24, bias=False)
    gemm(4096, 1280, 5120, residual=True)
