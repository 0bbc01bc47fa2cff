import sys; sys.path.insert(0, "/root/repo")
import torch
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda")
B, H, T, d = 16, 8, 4096, 40
C = H * d
q, k, v = (torch.randn(B * T, C, device=dev).half() for _ in range(3))
vt = ops.transpose_btc(v, B, T, C)
for _ in range(3):
    o, lse = ops.attn_fwd(q, k, vt, B, H, T, T, d, 1, need_lse=True)
do = torch.randn_like(o)
for _ in range(2):
    ops.attn_bwd(q, k, v, o, do, lse, B, H, T, T, d, 1)
torch.cuda.synchronize()
