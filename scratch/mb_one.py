import sys; sys.path.insert(0, "/root/repo")
import torch
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda")
B, H, Cin, Cout = 16, 64, 320, 320
x = torch.randn(B * H * H, Cin, device=dev).half()
w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).half()
bias = torch.randn(Cout, device=dev)
for _ in range(5):
    ops.conv3x3(x, w, B, H, H, bias=bias)
torch.cuda.synchronize()
