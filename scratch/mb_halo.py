"""Round 6: the halo-staged 3x3 convolution (csrc/gemm_halo.hip) against the per-tap ping-pong kernel (csrc/gemm_pp.hip) on the step's shapes, bench-hooks library
(FD_CONV_HALO=0 / 1 re-read per call): bit-equality of the outputs, then cold (rotating pool beyond the Infinity Cache) and hot timings of both.
usage: python scratch/mb_halo.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("FAIRDIFF_LIB", os.path.join(ROOT, "finetune_fair_diffusion_amd", "libfairdiff_hip_bench.so"))
sys.path.insert(0, ROOT)
import torch
from finetune_fair_diffusion_amd import lib, ops
dev = torch.device("cuda")


def t(fn, n=24):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def kname(B, H, Cin, Cout):
    d = lib.GemmDesc(); d.M, d.N, d.K, d.batch, d.conv, d.conv_mode, d.Bn, d.H, d.W, d.Cin, d.Ho, d.Wo = B * H * H, Cout, 9 * Cin, 1, 1, 0, B, H, H, Cin, H, H
    ws = ops.gemm_workspace(); d.workspace, d.workspace_bytes, d.ldc, d.lda, d.ldb = ws.data_ptr(), ws.numel() * 4, Cout, Cin, 9 * Cin
    buf = ctypes.create_string_buffer(128)
    lib.get().fd_gemm_kernel_name(ctypes.byref(d), buf, 128)
    return buf.value.decode()


shapes = [(16, 64, 320, 320), (16, 64, 640, 320), (16, 64, 960, 320), (16, 64, 640, 640), (16, 32, 640, 640), (16, 32, 1280, 640), (16, 32, 1920, 640), (16, 32, 1280, 1280),
          (16, 16, 1280, 1280), (16, 16, 2560, 1280), (8, 64, 320, 320), (8, 32, 640, 640), (8, 16, 1280, 1280)]
for (B, H, Cin, Cout) in shapes:
    M = B * H * H
    nset = max(2, min(16, int(600e6 / (M * (Cin + Cout) * 2))))
    xs = [torch.randn(M, Cin, device=dev).half() for _ in range(nset)]
    outs = [torch.empty(M, Cout, device=dev, dtype=torch.float16) for _ in range(nset)]
    w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).half(); bias = torch.randn(Cout, device=dev)
    res = {}
    for halo in ("0", "1"):
        os.environ["FD_CONV_HALO"] = halo
        name = kname(B, H, Cin, Cout)
        y = ops.conv3x3(xs[0], w, B, H, H, bias=bias)[0].clone()
        st = {"i": 0}
        def cold():
            i = st["i"] = (st["i"] + 1) % nset
            ops.conv3x3(xs[i], w, B, H, H, bias=bias, out=outs[i])
        us_c = t(cold)
        us_h = t(lambda: ops.conv3x3(xs[0], w, B, H, H, bias=bias, out=outs[0]))
        res[halo] = (name, y, us_c, us_h)
    fl = 2.0 * M * Cout * 9 * Cin
    (n0, y0, c0, h0), (n1, y1, c1, h1) = res["0"], res["1"]
    print(f"conv {Cin:4d}->{Cout:4d} @{H}^2 B{B}: {n0:34s} cold {c0:7.1f} hot {h0:7.1f} us | {n1:36s} cold {c1:7.1f} hot {h1:7.1f} us ({fl / c1 / 1e6:5.0f} TF) | cold x{c0 / c1:.3f} hot x{h0 / h1:.3f} | "
          f"bit-equal {bool(torch.equal(y0, y1))} maxdiff {float((y0.float() - y1.float()).abs().max()):.3e}", flush=True)
    del xs, outs
