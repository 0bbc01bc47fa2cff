"""Diagnostic: SD-v1.5-size VAE decode backward vs oracle: sensitivity to the gradient scale and to the upstream gradient's structure."""
import os, sys, time, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from finetune_fair_diffusion_amd import weights as W
from finetune_fair_diffusion_amd.vae import AutoencoderKL
from oracle import nn_vae
torch.set_num_threads(64)
dev = torch.device("cuda:0")
sd = W.synthetic_state_dict(W.vae_param_shapes(W.VAEConfig()), seed=2)
sd = {k: (v.half().float() if v.is_floating_point() else v) for k, v in sd.items()}
vo = nn_vae.AutoencoderKLDecoder(nn_vae.VAEConfig()); vo.load_state_dict(sd, strict=True); vo.requires_grad_(False)
vp = AutoencoderKL(W.VAEConfig(), sd, dev)
z = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(4))
zr = z.clone().requires_grad_(True)
pre = vo.decode(zr).sample
img_o = pre.clamp(-1, 1)
print("oracle: fraction of clamped pixels", float(((pre < -1) | (pre > 1)).float().mean()), " |pre| max", float(pre.abs().max()))
gens = {"randn*1e-3": torch.randn(img_o.shape, generator=torch.Generator().manual_seed(5)) * 1e-3,
        "smooth": (torch.linspace(-1, 1, 512)[None, None, :, None] * torch.linspace(1, -1, 512)[None, None, None, :]).expand(1, 3, 512, 512).contiguous() * 1e-3}
for name, g in gens.items():
    zr.grad = None
    (vo.decode(zr).sample.clamp(-1, 1) * g).sum().backward()
    ref = zr.grad.clone()
    for gs in (2.0 ** 8, 2.0 ** 11, 2.0 ** 14, 2.0 ** 16):
        img_p = vp.decode_images(z.to(dev), record=True)
        dz = vp.backward_images(g.to(dev), gs).cpu()
        cos = float(F.cosine_similarity(dz.flatten().double(), ref.flatten().double(), dim=0))
        print(f"{name:12s} gscale 2^{int(torch.log2(torch.tensor(gs)))}: rel max err {float((dz - ref).abs().max() / ref.abs().max()):.3e}  cosine {cos:.5f}  norm ratio {float(dz.norm() / ref.norm()):.4f}"
              f"  img err {float((img_p.float().cpu() - img_o).abs().max()):.3e}  finite {bool(torch.isfinite(dz).all())}")
# mask-flip effect alone: oracle gradient with the PRODUCT's clamp mask
pre_p = None
