"""A/B of the phased (two wave groups in opposite phases) main loop of the 8-wave GEMM / conv kernels: product library vs a bench-hooks build
made with EXTRA=-DFD_GEMM_PHASED.  Each library runs in its own process; correctness against fp32 torch on the same inputs."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    if sys.argv[1] != "base":
        os.environ["FAIRDIFF_LIB"] = os.path.join(ROOT, "scratch", f"libph_{sys.argv[1]}.so")
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
    g = torch.Generator(device="cuda").manual_seed(1)
    for (B, H, Cin, Cout) in [(16, 64, 320, 320), (16, 64, 640, 320), (16, 64, 960, 320), (8, 64, 320, 320)]:
        x = torch.randn(B * H * H, Cin, device=dev, generator=g).half(); w = (torch.randn(Cout, 9 * Cin, device=dev, generator=g) * 0.02).half(); bias = torch.randn(Cout, device=dev, generator=g)
        y, _, _ = ops.conv3x3(x, w, B, H, H, bias=bias)
        err = -1.0
        if B * H * H * Cin <= 16 * 32 * 32 * 1280:
            ref = F.conv2d(x.float().view(B, H, H, Cin).permute(0, 3, 1, 2), w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), bias, padding=1)
            err = float((y.float().view(B, H, H, Cout).permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max())
        us = t(lambda: ops.conv3x3(x, w, B, H, H, bias=bias))
        print(f"{sys.argv[1]:7s} conv {Cin:4d}->{Cout:4d} @{H}^2 b{B:<2d} {ops.gemm_kernel_name_last() if hasattr(ops, 'gemm_kernel_name_last') else '':s} {us:8.1f} us {2.0 * B * H * H * Cout * 9 * Cin / us / 1e6:8.1f} TF  err {err:.1e}")
else:
    for which in ["base"] + sorted(f[6:-3] for f in os.listdir(os.path.join(ROOT, "scratch")) if f.startswith("libph_") and f.endswith(".so")):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), which], capture_output=True, text=True)
        print(r.stdout, r.stderr[-800:] if r.returncode else "")
