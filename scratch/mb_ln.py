"""LayerNorm forward / backward timing and an output checksum (two builds are compared across processes: the shipped R-rows-per-wave kernel and the
one-row-per-wave form, bench-hooks library built with -DFD_LN_ONE_ROW; equal checksums = bit-identical outputs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


def chk(t):
    return int(t.contiguous().view(torch.int16).long().sum())


for M, C in ((32768, 320), (65536, 320), (8192, 640), (16384, 640), (2048, 1280), (4096, 1280), (2112, 1280), (2056, 768), (26, 768), (1001, 2048)):
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g).to(dev).half()
    dy = torch.randn(M, C, generator=g).to(dev).half()
    add = torch.randn(M, C, generator=g).to(dev).half()
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    y, st = ops.layernorm(x, gamma, beta, 1e-5, save_stats=True)
    dx = ops.layernorm_bwd(x, dy, gamma, st, add=add)
    tf = timeit(lambda: ops.layernorm(x, gamma, beta, 1e-5, save_stats=True))
    tb = timeit(lambda: ops.layernorm_bwd(x, dy, gamma, st, add=add))
    gb_f, gb_b = 2 * M * C * 2 / 1e9, 4 * M * C * 2 / 1e9
    print(f"LN {M}x{C}: fwd {tf:6.1f} us ({gb_f / tf * 1e6 / 1e3:5.2f} TB/s)  bwd {tb:6.1f} us ({gb_b / tb * 1e6 / 1e3:5.2f} TB/s)  checksums y {chk(y)} stats {chk(st.view(-1).view(torch.int16) if False else st.float().view(torch.int32).view(-1).to(torch.int16))} dx {chk(dx)}", flush=True)
