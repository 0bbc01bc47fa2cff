"""Does a LayerNorm row's result depend on WHERE the row sits (its index modulo the rows-per-wave count, or the total row count)?  It must not:
the shipped-vs-reference schedule test needs per-row results that are identical however the rows are batched."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda:0")
for M, C in ((4096, 320), (1024, 640), (512, 1280), (514, 768)):
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g).to(dev).half()
    dy = torch.randn(M, C, generator=g).to(dev).half()
    add = torch.randn(M, C, generator=g).to(dev).half()
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    y, st = ops.layernorm(x, gamma, beta, 1e-5, save_stats=True)
    dx = ops.layernorm_bwd(x, dy, gamma, st, add=add)
    for sh in (1, 2, 3):
        y2, st2 = ops.layernorm(x[sh:].contiguous(), gamma, beta, 1e-5, save_stats=True)
        dx2 = ops.layernorm_bwd(x[sh:].contiguous(), dy[sh:].contiguous(), gamma, st2, add=add[sh:].contiguous())
        print(f"{M}x{C} shift {sh}: fwd differing {int((y2 != y[sh:]).sum())}  stats differing {int((st2 != st[sh:]).sum())}  bwd differing {int((dx2 != dx[sh:]).sum())}")
    xx, dd, aa = torch.cat([x, x]), torch.cat([dy, dy]), torch.cat([add, add])
    y3, st3 = ops.layernorm(xx, gamma, beta, 1e-5, save_stats=True)
    dx3 = ops.layernorm_bwd(xx, dd, gamma, st3, add=aa)
    print(f"{M}x{C} doubled: fwd differing {int((y3[:M] != y).sum()) + int((y3[M:] != y).sum())}  bwd differing {int((dx3[:M] != dx).sum()) + int((dx3[M:] != dx).sum())}")
