"""fp32 PyTorch restatement of diffusers==0.19.3 ``AutoencoderKL.decode`` (SD-v1.5
VAE decoder).  TEST ORACLE -- parity unpinned (see oracle/__init__.py).

Reference call sites: exp-1-debias-gender/1-main-debias.py:730-733 (load),
:1058-1059 and :1133-1134 (``vae.decode(latents).sample``).  Parameter names follow
the diffusers layout (``post_quant_conv``, ``decoder.*``; mid-block attention uses
the 0.19 names ``group_norm,to_q,to_k,to_v,to_out.0``).
"""
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .nn_unet import ResnetBlock2D, Upsample2D


@dataclass
class VAEConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215


class VAEAttention(nn.Module):
    """Single-head spatial self-attention with residual (diffusers ``Attention`` built
    by ``UNetMidBlock2D`` with ``residual_connection=True, upcast_softmax=True``)."""

    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q = nn.Linear(c, c)
        self.to_k = nn.Linear(c, c)
        self.to_v = nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        B, C, H, W = x.shape
        res = x
        h = self.group_norm(x).reshape(B, C, H * W).transpose(1, 2)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        p = (torch.matmul(q, k.transpose(1, 2)) * (C ** -0.5)).float().softmax(dim=-1).to(q.dtype)
        o = self.to_out[0](torch.matmul(p, v))
        return o.transpose(1, 2).reshape(B, C, H, W) + res


class VAEMid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, None, groups, eps=1e-6), ResnetBlock2D(c, c, None, groups, eps=1e-6)])
        self.attentions = nn.ModuleList([VAEAttention(c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class UpDecoderBlock(nn.Module):
    def __init__(self, cin, cout, n, groups, add_up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, eps=1e-6) for i in range(n)])
        self.add_up = add_up
        if add_up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.add_up:
            x = self.upsamplers[0](x)
        return x


class Decoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.latent_channels, boc[-1], 3, padding=1)
        self.mid_block = VAEMid(boc[-1], g)
        rev = list(reversed(boc))
        ups, cout = [], rev[0]
        for i in range(len(rev)):
            cin, cout = cout, rev[i]
            ups.append(UpDecoderBlock(cin, cout, cfg.layers_per_block + 1, g, add_up=(i != len(rev) - 1)))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class AutoencoderKLDecoder(nn.Module):
    """Decoder half of ``AutoencoderKL`` (the encoder is never used on the path)."""

    def __init__(self, cfg: VAEConfig = VAEConfig()):
        super().__init__()
        self.config = cfg
        self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)
        self.decoder = Decoder(cfg)

    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype

    def decode(self, z):
        return DecoderOutput(self.decoder(self.post_quant_conv(z)))
