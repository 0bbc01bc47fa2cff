import sys; sys.path.insert(0, "/root/repo")
import torch
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda")
M, N, K = 65536, 2560, 320
a = [torch.randn(M, K, device=dev).half() for _ in range(4)]
b = (torch.randn(N, K, device=dev) * 0.02).half()
bias = torch.randn(N, device=dev)
outs = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(2)]
for i in range(12):
    ops.gemm(a[i % 4], b, bias=bias, out=outs[i % 2])
torch.cuda.synchronize()
