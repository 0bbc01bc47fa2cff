import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda:0")
os.environ["FD_GEMM_RB"] = "1"
M, N, K = 96 * 256, 320, 960
nb = K // 32
ramp = torch.zeros(K, device=dev)
for kb in range(nb):
    ramp[kb * 32:(kb + 1) * 32] = 2.0 ** (kb % 10) * (1 + kb // 10)      # distinct power-of-two-ish weights: a missing / swapped block is identifiable
tot = float(sum(2.0 ** (kb % 10) * (1 + kb // 10) for kb in range(nb)))
for name, a, b in (("A=ramp,B=1/32", ramp.half().expand(M, K).contiguous(), torch.full((N, K), 1 / 32, device=dev, dtype=torch.float16)),
                   ("A=1,B=ramp/32", torch.ones(M, K, device=dev, dtype=torch.float16), (ramp / 32).half().expand(N, K).contiguous())):
    o = ops.gemm(a, b).float()
    torch.cuda.synchronize()
    bad = o != tot
    print(f"{name}: expected {tot}: bad {int(bad.sum())}")
    d = (o[bad] - tot)
    vals, cnt = torch.unique(d, return_counts=True)
    print("   deltas (value: count):", sorted(zip(cnt.tolist(), vals.tolist()), reverse=True)[:12])
    rows = bad.any(1).nonzero().flatten()
    cols = bad.any(0).nonzero().flatten()
    print("   rows mod 96:", sorted(set((rows % 96).tolist()))[:40], " cols mod 80:", sorted(set((cols % 80).tolist()))[:40])
