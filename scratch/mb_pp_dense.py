"""Round 6: the dense GEMMs of the step on the ping-pong kernel (csrc/gemm_pp.hip, operands by buffer_load ... lds since this round) against the shipped
tile policy (lockstep gemm_big_kernel for everything but FF1 at 64^2), bench-hooks library (FD_GEMM_PP re-read per call): cold operands (rotating pool beyond the
Infinity Cache), equality of the results, time per launch.     usage: python scratch/mb_pp_dense.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("FAIRDIFF_LIB", os.path.join(ROOT, "finetune_fair_diffusion_amd", "libfairdiff_hip_bench.so"))
sys.path.insert(0, ROOT)
import torch
from finetune_fair_diffusion_amd import lib, ops
dev = torch.device("cuda")
BASE = 1 | 4 | 8 | 32
MODES = [("shipped", BASE), ("pp256", BASE | 2), ("pp256+128", BASE | 2 | 16)]


def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def kname(M, N, K, K2):
    d = lib.GemmDesc(); d.M, d.N, d.K, d.K2, d.batch, d.ldc, d.lda, d.ldb = M, N, K, K2, 1, N, K, K
    ws = ops.gemm_workspace(); d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    if K2:
        d.A2 = d.B2 = ws.data_ptr(); d.lda2 = d.ldb2 = K2
    buf = ctypes.create_string_buffer(128)
    lib.get().fd_gemm_kernel_name(ctypes.byref(d), buf, 128)
    return buf.value.decode().replace("kernel", "")


shapes = [(65536, 320, 320, 0, 0), (65536, 320, 320, 8, 0), (65536, 320, 320, 0, 1), (65536, 960, 320, 24, 0), (65536, 320, 1280, 0, 1), (65536, 320, 2560, 0, 0), (65536, 1280, 320, 0, 0),
          (16384, 640, 640, 0, 0), (16384, 640, 640, 8, 0), (16384, 640, 640, 0, 1), (16384, 1920, 640, 24, 0), (16384, 640, 2560, 0, 1), (16384, 640, 5120, 0, 0), (16384, 2560, 640, 0, 0),
          (4096, 1280, 1280, 0, 0), (4096, 1280, 1280, 8, 0), (4096, 1280, 1280, 0, 1), (4096, 3840, 1280, 24, 0), (4096, 1280, 5120, 0, 1), (4096, 1280, 10240, 0, 0), (4096, 5120, 1280, 0, 0),
          (4000, 1280, 1288, 8, 1)]
for (M, N, K, K2, R) in shapes:
    nset = max(2, min(12, int(500e6 / ((M * K + M * N * (1 + R)) * 2))))
    As = [torch.randn(M, K, device=dev).half() for _ in range(nset)]
    b = (torch.randn(N, K, device=dev) * K ** -0.5).half(); bias = torch.randn(N, device=dev)
    a2 = torch.randn(M, K2, device=dev).half() if K2 else None
    b2 = (torch.randn(N, K2, device=dev) * 0.1).half() if K2 else None
    res = [torch.randn(M, N, device=dev).half() for _ in range(nset)] if R else None
    outs = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(nset)]
    line, ys = f"gemm {M:6d}x{N:5d}x{K:5d}+{K2:2d}{' +res' if R else '     '}: ", []
    for name, mode in MODES:
        os.environ["FD_GEMM_PP"] = str(mode)
        kn = kname(M, N, K, K2)
        ys.append(ops.gemm(As[0], b, a2=a2, b2=b2, bias=bias, residual=res[0] if R else None).clone())
        st = {"i": 0}
        def cold():
            i = st["i"] = (st["i"] + 1) % nset
            ops.gemm(As[i], b, a2=a2, b2=b2, bias=bias, residual=res[i] if R else None, out=outs[i])
        us = t(cold)
        line += f"{name} {kn:26s} {us:7.1f} us ({2.0 * M * N * (K + K2) / us / 1e6:5.0f} TF) | "
    ref = (As[0].float() @ b.float().t() + (a2.float() @ b2.float().t() if K2 else 0) + bias + (res[0].float() if R else 0))
    err = float((ys[-1].float() - ref).abs().max() / ref.abs().max())
    print(line + f"equal {bool(torch.equal(ys[0], ys[1]))} {bool(torch.equal(ys[0], ys[2]))}  err vs fp32 {err:.1e}", flush=True)
    del As, outs, res
