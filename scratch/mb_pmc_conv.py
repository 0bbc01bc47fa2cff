"""PMC target: the dominant kernel of the bench step, gemm_big_kernel<256,320,conv3x3>, on its three U-Net shapes
(64x64 level, CFG batch 16, Cout 320, Cin 320/640/960).  Buffers are rotated over 8 sets (> 256 MiB Infinity Cache)
so that FETCH_SIZE reflects HBM traffic rather than on-die re-use of a hot microbenchmark buffer."""
import sys; sys.path.insert(0, "/root/repo")
import torch
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda")
B, H, Cout = 16, 64, 320
for Cin in (320, 640, 960):
    xs = [torch.randn(B * H * H, Cin, device=dev).half() for _ in range(8)]
    w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).half()
    bias = torch.randn(Cout, device=dev)
    outs = [torch.empty(B * H * H, Cout, device=dev, dtype=torch.float16) for _ in range(8)]
    for i in range(16):
        ops.conv3x3(xs[i % 8], w, B, H, H, bias=bias, out=outs[i % 8])
    torch.cuda.synchronize()
print("done")
