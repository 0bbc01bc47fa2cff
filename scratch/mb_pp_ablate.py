"""Main-loop ablation of the ping-pong convolution / FF1 kernels (csrc/gemm_pp.hip) on the step's shapes, bench-hooks library, FD_GEMM_DBG:
0 = the kernel, 1 = DMA from the zero page after the prologue (no operand traffic), 2 = no MFMAs (fragment reads + waits + barriers + DMA), 3 = no epilogue,
6 = no fragment reads (MFMAs on stale registers), 7 = no vmcnt waits / barriers.  Operands cold (rotating pool) or hot.  One process per mode (the switch is read once).
usage: python scratch/mb_pp_ablate.py [mode]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    os.environ["FD_GEMM_DBG"] = sys.argv[1]
    os.environ["FAIRDIFF_LIB"] = os.path.join(ROOT, "finetune_fair_diffusion_amd", "libfairdiff_hip_bench.so")
    sys.path.insert(0, ROOT)
    import torch
    from finetune_fair_diffusion_amd import ops
    dev = torch.device("cuda")
    def t(fn, n=24):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    for (B, H, Cin, Cout) in [(16, 64, 320, 320), (16, 64, 640, 320), (16, 32, 640, 640), (16, 32, 1280, 640), (16, 16, 1280, 1280)]:
        M = B * H * H
        nset = max(2, min(16, int(600e6 / (M * (Cin + Cout) * 2))))
        xs = [torch.randn(M, Cin, device=dev).half() for _ in range(nset)]
        outs = [torch.empty(M, Cout, device=dev, dtype=torch.float16) for _ in range(nset)]
        w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).half(); bias = torch.randn(Cout, device=dev)
        st = {"i": 0}
        def cold():
            i = st["i"] = (st["i"] + 1) % nset
            ops.conv3x3(xs[i], w, B, H, H, bias=bias, out=outs[i])
        us_c = t(cold)
        us_h = t(lambda: ops.conv3x3(xs[0], w, B, H, H, bias=bias, out=outs[0]))
        fl = 2.0 * M * Cout * 9 * Cin
        print(f"DBG={sys.argv[1]}  conv {Cin}->{Cout} @{H}^2   cold {us_c:8.1f} us ({fl / us_c / 1e6:7.0f} TF nominal)   hot {us_h:8.1f} us ({fl / us_h / 1e6:7.0f})", flush=True)
        del xs, outs
else:
    for dbg in ("0", "1", "2", "6", "7", "3"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), dbg], capture_output=True, text=True)
        print(r.stdout, r.stderr[-800:] if r.returncode else "")
