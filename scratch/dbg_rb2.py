import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda:0")
os.environ["FD_GEMM_RB"] = "1"
M, N, K = 96 * 256, 320, 960
for blk in (0, 5, 12, 20, 29):
    a = torch.zeros(M, K, device=dev, dtype=torch.float16)
    for kb in range(30):
        a[:, kb * 32:(kb + 1) * 32] = 1 + kb
    b = torch.zeros(N, K, device=dev, dtype=torch.float16)
    b[:, blk * 32:(blk + 1) * 32] = 1.0 / 32
    o = ops.gemm(a, b).float()
    torch.cuda.synchronize()
    bad = (o != 1 + blk)
    print(f"B block {blk}: bad {int(bad.sum())}")
    rows = bad.any(1).nonzero().flatten().tolist()
    # group rows into tiles
    tiles = sorted(set(r // 96 for r in rows))
    for t in tiles[:6]:
        sub = bad[t * 96:(t + 1) * 96]
        rr = sub.any(1).nonzero().flatten().tolist()
        cc = sub.any(0).nonzero().flatten().tolist()
        vals = torch.unique(o[t * 96:(t + 1) * 96][sub]).tolist()
        print(f"   tile {t}: rows {rr[0]}..{rr[-1]} (n={len(rr)}) cols {cc[0]}..{cc[-1]} (n={len(cc)}) count {int(sub.sum())} values {vals[:6]}")
