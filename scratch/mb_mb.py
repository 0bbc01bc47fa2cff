"""A/B of an experimental GEMM kernel family (now: gemm_p4.hip, FD_GEMM_P4=1; round 2 first used it for scratch/gemm_mb_experiment.hip) against the
current tile policy, on the step's shapes, with a correctness check.
Bench-hooks library; each arm in its own process (the switch is read once).  Usage: python scratch/mb_mb.py [arm]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    if sys.argv[1] != "base":
        os.environ["FD_GEMM_P4"] = "1"
    os.environ["FAIRDIFF_LIB"] = os.path.join(ROOT, "finetune_fair_diffusion_amd", "libfairdiff_hip_bench.so")
    sys.path.insert(0, ROOT)
    import torch
    import torch.nn.functional as F
    from finetune_fair_diffusion_amd import ops
    dev = torch.device("cuda")
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    for (B, H, Cin, Cout) in [(16, 64, 320, 320), (16, 64, 640, 320), (16, 32, 640, 640), (16, 32, 1280, 640), (16, 16, 1280, 1280), (16, 16, 2560, 1280), (16, 8, 1280, 1280), (8, 64, 320, 320)]:
        x = torch.randn(B * H * H, Cin, device=dev).half(); w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).half(); bias = torch.randn(Cout, device=dev)
        res = torch.randn(B * H * H, Cout, device=dev).half()
        y, _, _ = ops.conv3x3(x, w, B, H, H, bias=bias, residual=res)
        err = -1.0
        if B * H * H <= 16384:
            ref = F.conv2d(x.float().reshape(B, H, H, Cin).permute(0, 3, 1, 2), w.float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout) + res.float()
            err = float((y.float() - ref).abs().max() / ref.abs().max())
        us = t(lambda: ops.conv3x3(x, w, B, H, H, bias=bias, residual=res))
        print(f"{sys.argv[1]:>5s}  conv {Cin:4d}->{Cout:4d} @{H:2d}^2 b{B:<2d}      {us:8.1f} us {2.0 * B * H * H * Cout * 9 * Cin / us / 1e6:8.1f} TF  err {err:.1e}")
    for (M, N, K, act) in [(65536, 320, 320, "none"), (65536, 960, 320, "none"), (65536, 320, 1280, "none"), (65536, 2560, 320, "geglu"), (32768, 320, 320, "none"),
                           (16384, 640, 640, "none"), (16384, 1920, 640, "none"), (16384, 640, 2560, "none"), (16384, 5120, 640, "geglu"),
                           (4096, 1280, 1280, "none"), (4096, 3840, 1280, "none"), (4096, 1280, 5120, "none"), (4096, 10240, 1280, "geglu"), (1024, 1280, 1280, "none")]:
        a = torch.randn(M, K, device=dev).half(); b = (torch.randn(N, K, device=dev) * 0.05).half()
        bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev).half() if act == "none" else None
        t8 = torch.randn(M, 8, device=dev).half(); up = (torch.randn(N, 8, device=dev) * 0.05).half()
        kw = dict(bias=bias, residual=res, act=act) if act == "none" else dict(bias=bias, act=act)
        if act == "none":
            kw.update(a2=t8, b2=up)
        y = ops.gemm(a, b, **kw)
        pre = a.float() @ b.float().t() + bias + (t8.float() @ up.float().t() if act == "none" else 0)
        ref = pre + res.float() if act == "none" else (pre.half().float()[:, 0::2] * F.gelu(pre.half().float()[:, 1::2]))
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        us = t(lambda: ops.gemm(a, b, **kw))
        print(f"{sys.argv[1]:>5s}  gemm {M:6d}x{N:5d}x{K:5d} {act:6s} {us:8.1f} us {2.0 * M * N * (K + (8 if act == 'none' else 0)) / us / 1e6:8.1f} TF  err {err:.1e}")
else:
    outs = {}
    for arm in ("base", "p4"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), arm], capture_output=True, text=True)
        outs[arm] = [l for l in r.stdout.splitlines() if l.strip()]
        if r.returncode:
            print(r.stderr[-1500:])
    for a, b in zip(outs["base"], outs["p4"]):
        print(a); print(b)
