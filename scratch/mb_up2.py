import sys; sys.path.insert(0, "/root/repo")
import torch
from finetune_fair_diffusion_amd import ops
from finetune_fair_diffusion_amd.layers import Conv3x3
dev = torch.device("cuda")
def bench(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
# U-Net upsamplers at CFG batch 16, VAE upsamplers at batch 8
for (B, H, C) in [(16, 32, 640), (16, 16, 1280), (16, 8, 1280), (8, 64, 512), (8, 128, 512), (8, 256, 256)]:
    w = (torch.randn(C, C, 3, 3) * 0.02).to(dev).half(); bias = torch.zeros(C, device=dev)
    conv = Conv3x3({"c.weight": w, "c.bias": bias}, "c", dev)
    x = torch.randn(B * H * H, C, device=dev).half()
    g = torch.randn(B * 4 * H * H, C, device=dev).half()
    conv.wk_up2p, conv.wd_up2p, conv.wd
    t_new = bench(lambda: ops.conv_up2(x, conv, B, H, H))
    t_old = bench(lambda: ops.conv3x3(x, conv.wk, B, H, H, mode=ops.CONV_UP2, bias=conv.bias))
    b_new = bench(lambda: ops.conv_up2_bwd(g, conv, B, H, H))
    b_old = bench(lambda: ops.downsum2x2(ops.conv3x3(g, conv.wd, B, 2 * H, 2 * H)[0], B, H, H, C))
    print(f"up2 conv {C}@{H}->{2*H} B={B}: fwd {t_new:8.1f} us (3x3 gather {t_old:8.1f})   dgrad {b_new:8.1f} us (3x3 + downsum {b_old:8.1f})")
