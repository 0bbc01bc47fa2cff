import sys; sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from finetune_fair_diffusion_amd import ops
dev=torch.device('cuda')
x=torch.linspace(-4,4,33,device=dev).half()
xr=x.float().requires_grad_(True)
F.hardswish(xr).backward(torch.ones_like(xr))
mine=ops.act_bwd(x, torch.ones_like(x), "hardswish")
for a,b,c in zip(x.tolist(), xr.grad.tolist(), mine.tolist()): print(a,b,c)
