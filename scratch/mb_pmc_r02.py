"""PMC target (round 2): the kernel the bench's roofline names -- gemm_big_kernel<128, 320, 4, 4, 1> (16-wave 128x320 implicit-GEMM conv) -- on
its U-Net shapes at CFG batch 16: 32x32 level (640 ch, no split-K), 16x16 level (1280 ch, split-K 2) and 8x8 level (1280 ch, split-K 8).
Buffers are rotated over enough sets to exceed the 256 MiB Infinity Cache so that FETCH_SIZE reflects memory-side traffic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda")
for (B, H, C) in ((16, 32, 640), (16, 16, 1280), (16, 8, 1280)):
    nset = max(8, int(300e6 / (B * H * H * C * 2 * 2)) + 1)
    xs = [torch.randn(B * H * H, C, device=dev).half() for _ in range(nset)]
    w = (torch.randn(C, 9 * C, device=dev) * 0.02).half()
    bias = torch.randn(C, device=dev)
    outs = [torch.empty(B * H * H, C, device=dev, dtype=torch.float16) for _ in range(nset)]
    for i in range(2 * nset):
        ops.conv3x3(xs[i % nset], w, B, H, H, bias=bias, out=outs[i % nset])
    torch.cuda.synchronize()
print("done")
