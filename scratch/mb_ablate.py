"""Main-loop ablation of the big-tile GEMM/conv kernels on the step's heaviest shapes (bench-hooks library, FD_GEMM_DBG):
0 = full kernel, 1 = no loads after the first k-tile (MFMA + LDS fragment reads only), 2 = no MFMAs (operand loads + barriers only),
3 = no epilogue.  Each configuration runs in a fresh process (the switch is read once).  Usage: python scratch/mb_ablate.py [dbg]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    os.environ["FD_GEMM_DBG"] = sys.argv[1]
    os.environ["FAIRDIFF_LIB"] = os.path.join(ROOT, "finetune_fair_diffusion_amd", "libfairdiff_hip_bench.so")
    sys.path.insert(0, ROOT)
    import torch
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
    rows = []
    for (B, H, Cin, Cout) in [(16, 64, 320, 320), (16, 32, 640, 640), (16, 16, 1280, 1280), (16, 8, 1280, 1280)]:
        x = torch.randn(B * H * H, Cin, device=dev).half(); w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).half(); bias = torch.randn(Cout, device=dev)
        us = t(lambda: ops.conv3x3(x, w, B, H, H, bias=bias))
        rows.append((f"conv {Cin}->{Cout} @{H}^2 b{B}", us, 2.0 * B * H * H * Cout * 9 * Cin / us / 1e6))
    for (M, N, K) in [(65536, 320, 320), (65536, 320, 1280), (65536, 2560, 320), (16384, 640, 640), (16384, 640, 2560), (4096, 1280, 1280), (4096, 1280, 5120), (1024, 1280, 1280)]:
        a = torch.randn(M, K, device=dev).half(); b = (torch.randn(N, K, device=dev) * 0.05).half()
        us = t(lambda: ops.gemm(a, b))
        rows.append((f"gemm {M}x{N}x{K}", us, 2.0 * M * N * K / us / 1e6))
    for name, us, tf in rows:
        print(f"DBG={sys.argv[1]}  {name:28s} {us:9.1f} us  {tf:8.1f} TFLOP/s(nominal)")
else:
    for dbg in ("0", "1", "2", "3"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), dbg], capture_output=True, text=True)
        print(r.stdout, r.stderr[-500:] if r.returncode else "")
