"""PMC target: gemm_big_kernel<256,320,dense> on its most frequent U-Net shapes (buffers rotated beyond the Infinity Cache)."""
import sys; sys.path.insert(0, "/root/repo")
import torch
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda")
for (M, N, K) in [(65536, 2560, 320), (65536, 320, 320), (65536, 320, 1280), (16384, 5120, 640), (4096, 10240, 1280)]:
    nb = max(2, int(600e6 // (M * K * 2 + M * N * 2)) + 1)
    xs = [torch.randn(M, K, device=dev).half() for _ in range(nb)]
    w = (torch.randn(N, K, device=dev) * 0.02).half()
    bias = torch.randn(N, device=dev)
    outs = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(nb)]
    for i in range(12):
        ops.gemm(xs[i % nb], w, bias=bias, out=outs[i % nb])
    torch.cuda.synchronize()
print("done")
