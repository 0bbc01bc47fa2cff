"""A/B of the register-B streaming GEMM (csrc/gemm_rb.hip) against the kernels the round-4 policy picks, on the step's dense shapes, with a bit-exactness check
(same k order and rounding: results must be EQUAL).  Operands are COLD: every launch reads another copy of A (and writes another C) out of a pool larger than
the 256 MB Infinity Cache, as in the step, where A was written by the previous kernel; the weights stay hot.  Bench-hooks library (make BENCH_HOOKS=1;
FAIRDIFF_LIB=.../libfairdiff_hip_bench.so): FD_GEMM_RB is re-read on every call there.      python scratch/mb_rb.py [quick]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
POOL = 768 << 20        # bytes of A copies (and of C copies) cycled through


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dev).half()


def gemm(M, N, K, residual=False, k2=0, act="none", bias=True, colscale=None, check_only=False):
    n_out = N // 2 if act == "geglu" else N
    ncopy = max(2, min(24, POOL // max(M * K * 2, M * n_out * 2)))
    a0 = rnd(M, K)
    As = [a0] + [a0.clone() for _ in range(ncopy - 1)]
    b = rnd(N, K, scale=K ** -0.5)
    bv = torch.randn(N, generator=g).to(dev) if bias else None
    res = rnd(M, n_out) if residual else None
    a2, b2 = (rnd(M, k2), rnd(N, k2, scale=0.1)) if k2 else (None, None)
    outs = [torch.empty(M, n_out, dtype=torch.float16, device=dev) for _ in range(ncopy)]
    state = {"i": 0}

    def fn():
        i = state["i"] = (state["i"] + 1) % ncopy
        return ops.gemm(As[i], b, a2=a2, b2=b2, bias=bv, residual=res, act=act, out=outs[i], colscale=colscale)

    res_o, ts = {}, {}
    for mode in ("0", "1"):
        os.environ["FD_GEMM_RB"] = mode
        res_o[mode] = fn().float().clone()
        torch.cuda.synchronize()
        if check_only:
            continue
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        n = 40
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts[mode] = 1e3 * e0.elapsed_time(e1) / n
    bad = int((res_o["0"] != res_o["1"]).sum())
    fl = 2.0 * M * N * (K + k2)
    by = 2.0 * (M * K + N * K + M * n_out * (2 if residual else 1))
    name = f"gemm {M}x{N}x{K}{'+' + str(k2) if k2 else ''}{' +res' if residual else ''}{' ' + act if act != 'none' else ''}{'' if bias else ' nobias'}{' colscale' if colscale else ''}"
    if check_only:
        print(f"{name:48s} differing outputs {bad}", flush=True)
    else:
        print(f"{name:48s} shipped {ts['0']:7.1f} us ({fl / ts['0'] / 1e6:6.0f} TF/s)   register-B {ts['1']:7.1f} us ({fl / ts['1'] / 1e6:6.0f} TF/s, {by / ts['1'] / 1e6:5.2f} TB/s)"
              f"   {ts['0'] / ts['1']:5.2f}x   differing outputs {bad}", flush=True)
    assert (bad == 0 or os.environ.get("MB_RB_NOASSERT")) and torch.isfinite(res_o["1"]).all(), name


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    # awkward sizes first: ragged M (not a multiple of 96), LoRA slabs of 8 / 24, full second slab, several column blocks, fewer tiles than workgroups
    gemm(96 * 300 + 40, 320, 320, check_only=True)
    gemm(128 * 257, 640, 320, residual=True, k2=8, check_only=True)
    gemm(128 * 200 + 8, 960, 320, k2=24, bias=False, colscale=(0.25, 320), check_only=True)
    gemm(96 * 513 + 1, 320, 640, residual=True, check_only=True)
    gemm(128 * 100, 2560, 320, act="geglu", check_only=True)
    gemm(4096, 1280, 1280, k2=1280, check_only=True)
    gemm(4096, 1280, 1280, k2=640, check_only=True)
    gemm(1000, 320, 320, k2=8, residual=True, check_only=True)
    gemm(16384, 640, 640, act="silu", check_only=True)
    if quick:
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "k320":
        for _ in range(2):
            gemm(65536, 320, 320)
            gemm(65536, 320, 320, residual=True)
            gemm(65536, 320, 320, k2=8)
            gemm(65536, 960, 320, k2=24, bias=False, colscale=(0.25, 320))
            gemm(65536, 960, 320)
            gemm(65536, 1280, 320)
            gemm(65536, 2560, 320, act="geglu")
            gemm(65536, 2560, 320)
            gemm(32768, 320, 320)
        sys.exit(0)
    # the step's dense shapes (profiles/r05_gemm_shapes_single_stream_start_of_round.csv)
    gemm(65536, 320, 320)
    gemm(65536, 320, 320, residual=True)
    gemm(65536, 320, 320, k2=8)
    gemm(65536, 320, 320, k2=8, residual=True)
    gemm(65536, 960, 320, k2=24, bias=False, colscale=(0.25, 320))
    gemm(65536, 1280, 320)
    gemm(65536, 2560, 320, act="geglu")
    gemm(65536, 320, 1280, residual=True)
    gemm(65536, 320, 2560)
    gemm(65536, 320, 960)
    gemm(16384, 640, 640)
    gemm(16384, 640, 640, k2=8, residual=True)
    gemm(16384, 1920, 640, k2=24, bias=False, colscale=(0.25, 640))
    gemm(16384, 2560, 640)
    gemm(16384, 5120, 640, act="geglu")
    gemm(16384, 640, 2560, residual=True)
    gemm(16384, 640, 5120)
    gemm(16384, 640, 1920)
    gemm(4096, 1280, 1280)
    gemm(4096, 1280, 1280, k2=8, residual=True)
    gemm(4096, 3840, 1280, k2=24, bias=False, colscale=(0.25, 1280))
    gemm(4096, 5120, 1280)
    gemm(4096, 10240, 1280, act="geglu")
    gemm(4096, 1280, 5120, residual=True)
    gemm(4096, 1280, 10240)
    gemm(4096, 1280, 3840)
    gemm(1024, 1280, 1280)
    gemm(1024, 1280, 5120, residual=True)
