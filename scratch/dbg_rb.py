import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
def rnd(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale).to(dev).half()
def run(M, N, K, k2=0, res=False, bias=True):
    a, b = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    r = rnd(M, N) if res else None
    bv = torch.randn(N, generator=g).to(dev) if bias else None
    a2, b2 = (rnd(M, k2), rnd(N, k2, scale=0.1)) if k2 else (None, None)
    o = {}
    for mode in ("0", "1"):
        os.environ["FD_GEMM_RB"] = mode
        o[mode] = ops.gemm(a, b, a2=a2, b2=b2, residual=r, bias=bv).float()
        torch.cuda.synchronize()
    d = (o["0"] != o["1"])
    rows = d.any(1).nonzero().flatten()
    cols = d.any(0).nonzero().flatten()
    print(f"M={M} N={N} K={K} k2={k2} res={res} bias={bias}: differing {int(d.sum())} maxdiff {float((o['0']-o['1']).abs().max()):.3e} rows {rows[:4].tolist()}..{rows[-3:].tolist() if len(rows) else []} (n={len(rows)}) cols n={len(cols)} {cols[:6].tolist()}", flush=True)
for sh in [(96*300+40, 320, 320), (96*300, 320, 320), (96*600, 320, 320)]:
    run(*sh)
    run(*sh, bias=False)
