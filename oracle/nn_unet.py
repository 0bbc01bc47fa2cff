"""fp32 PyTorch restatement of diffusers==0.19.3 ``UNet2DConditionModel`` with
``LoRAAttnProcessor`` (TEST ORACLE -- parity unpinned, see oracle/__init__.py).

Call sites in the reference: exp-1-debias-gender/1-main-debias.py:734-737 (load),
:798-818 (LoRA injection), :1046-1050 and :1118-1122 (forward).  The arithmetic
lives in the un-vendored diffusers package (environment.yml:128); module and
parameter names below follow the diffusers state-dict layout so SD-v1.5 weights
and the reference's exported ``unet_lora.pth`` (2-export-checkpoint.py:630-634)
load unchanged.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    # SD-v1.5 stores the *number of heads* under ``attention_head_dim`` (=8)
    attention_head_dim: int = 8
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D",
                                         "CrossAttnDownBlock2D", "DownBlock2D")
    up_block_types: Tuple[str, ...] = ("UpBlock2D", "CrossAttnUpBlock2D",
                                       "CrossAttnUpBlock2D", "CrossAttnUpBlock2D")
    sample_size: int = 64


def timestep_embedding(timesteps: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers ``get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)``."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


class LoRALinearLayer(nn.Module):
    """diffusers ``LoRALinearLayer``: fp32 down/up, N(0,1/r) / zeros init."""

    def __init__(self, in_features, out_features, rank=4):
        super().__init__()
        self.down = nn.Linear(in_features, rank, bias=False)
        self.up = nn.Linear(rank, out_features, bias=False)
        nn.init.normal_(self.down.weight, std=1 / rank)
        nn.init.zeros_(self.up.weight)

    def forward(self, x):
        orig = x.dtype
        return self.up(self.down(x.to(self.down.weight.dtype))).to(orig)


class LoRAAttnProcessor(nn.Module):
    def __init__(self, hidden_size, cross_attention_dim=None, rank=4):
        super().__init__()
        self.hidden_size, self.cross_attention_dim, self.rank = hidden_size, cross_attention_dim, rank
        self.to_q_lora = LoRALinearLayer(hidden_size, hidden_size, rank)
        self.to_k_lora = LoRALinearLayer(cross_attention_dim or hidden_size, hidden_size, rank)
        self.to_v_lora = LoRALinearLayer(cross_attention_dim or hidden_size, hidden_size, rank)
        self.to_out_lora = LoRALinearLayer(hidden_size, hidden_size, rank)


class Attention(nn.Module):
    """diffusers ``Attention`` (cross_attention) with optional LoRA processor."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, bias=False):
        super().__init__()
        self.heads = heads
        self.scale = (query_dim // heads) ** -0.5
        kv = cross_attention_dim or query_dim
        self.to_q = nn.Linear(query_dim, query_dim, bias=bias)
        self.to_k = nn.Linear(kv, query_dim, bias=bias)
        self.to_v = nn.Linear(kv, query_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(query_dim, query_dim), nn.Dropout(0.0)])
        self.processor: Optional[LoRAAttnProcessor] = None  # not a registered submodule name clash
        self.softmax_dtype = None  # None -> input dtype (upcast_softmax=False)

    def forward(self, hidden_states, encoder_hidden_states=None):
        p = self.processor
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q = self.to_q(hidden_states)
        k = self.to_k(ctx)
        v = self.to_v(ctx)
        if p is not None:
            q = q + p.to_q_lora(hidden_states)
            k = k + p.to_k_lora(ctx)
            v = v + p.to_v_lora(ctx)
        B, T, C = q.shape
        h = self.heads
        def split(x):
            return x.reshape(B, x.shape[1], h, C // h).permute(0, 2, 1, 3)
        q, k, v = split(q), split(k), split(v)
        scores = torch.matmul(q, k.transpose(-1, -2)) * self.scale
        probs = scores.softmax(dim=-1)
        out = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(B, T, C)
        o = self.to_out[0](out)
        if p is not None:
            o = o + p.to_out_lora(out)
        return o


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_attention_dim, heads)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, enc):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), enc) + x
        x = self.ff(self.norm3(x)) + x
        return x


class Transformer2DModel(nn.Module):
    def __init__(self, channels, heads, cross_attention_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, channels, eps=1e-6)
        self.proj_in = nn.Conv2d(channels, channels, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(channels, heads, cross_attention_dim)])
        self.proj_out = nn.Conv2d(channels, channels, 1)

    def forward(self, x, enc):
        B, C, H, W = x.shape
        res = x
        h = self.proj_in(self.norm(x))
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, enc)
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        return self.proj_out(h) + res


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_channels=1280, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, cout) if temb_channels else None
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, n, attn, heads, xdim, groups, add_down, temb_dim):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb_dim, groups=groups) for i in range(n)])
        if attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, xdim, groups) for _ in range(n)])
        self.has_attn = attn
        if add_down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout)])
        self.add_down = add_down

    def forward(self, x, temb, enc):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.has_attn:
                x = self.attentions[i](x, enc)
            outs.append(x)
        if self.add_down:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, c, heads, xdim, groups, temb_dim):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb_dim, groups=groups), ResnetBlock2D(c, c, temb_dim, groups=groups)])
        self.attentions = nn.ModuleList([Transformer2DModel(c, heads, xdim, groups)])

    def forward(self, x, temb, enc):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, enc)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, cin, cout, cprev, n, attn, heads, xdim, groups, add_up, temb_dim):
        super().__init__()
        rs = []
        for i in range(n):
            skip = cin if i == n - 1 else cout
            rin = cprev if i == 0 else cout
            rs.append(ResnetBlock2D(rin + skip, cout, temb_dim, groups=groups))
        self.resnets = nn.ModuleList(rs)
        if attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, xdim, groups) for _ in range(n)])
        self.has_attn = attn
        if add_up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])
        self.add_up = add_up

    def forward(self, x, skips, temb, enc):
        for i, r in enumerate(self.resnets):
            x = torch.cat([x, skips.pop()], dim=1)
            x = r(x, temb)
            if self.has_attn:
                x = self.attentions[i](x, enc)
        if self.add_up:
            x = self.upsamplers[0](x)
        return x


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.linear_1 = nn.Linear(cin, cout)
        self.linear_2 = nn.Linear(cout, cout)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class UNetOutput:
    def __init__(self, sample):
        self.sample = sample


class UNet2DConditionModel(nn.Module):
    def __init__(self, cfg: UNetConfig = UNetConfig()):
        super().__init__()
        self.config = cfg
        boc = cfg.block_out_channels
        g, heads, xdim = cfg.norm_num_groups, cfg.attention_head_dim, cfg.cross_attention_dim
        temb_dim = boc[0] * 4
        self.temb_dim = temb_dim
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb_dim)
        downs = []
        cout = boc[0]
        for i, t in enumerate(cfg.down_block_types):
            cin, cout = cout, boc[i]
            downs.append(DownBlock(cin, cout, cfg.layers_per_block, t.startswith("CrossAttn"), heads, xdim, g,
                                   add_down=(i != len(boc) - 1), temb_dim=temb_dim))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = MidBlock(boc[-1], heads, xdim, g, temb_dim)
        ups = []
        rev = list(reversed(boc))
        cout = rev[0]
        for i, t in enumerate(cfg.up_block_types):
            cprev, cout = cout, rev[i]
            cin = rev[min(i + 1, len(boc) - 1)]
            ups.append(UpBlock(cin, cout, cprev, cfg.layers_per_block + 1, t.startswith("CrossAttn"), heads, xdim, g,
                               add_up=(i != len(boc) - 1), temb_dim=temb_dim))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-5)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    # ---- diffusers-compatible attention-processor surface -------------------
    def _attn_modules(self) -> Dict[str, Attention]:
        out = {}
        for name, m in self.named_modules():
            if isinstance(m, Attention):
                out[name + ".processor"] = m
        return out

    @property
    def attn_processors(self):
        return {k: m.processor for k, m in self._attn_modules().items()}

    def set_attn_processor(self, procs: Dict[str, LoRAAttnProcessor]):
        mods = self._attn_modules()
        assert set(procs) == set(mods), "processor dict keys must match attention layers"
        for k, p in procs.items():
            mods[k].processor = p

    def forward(self, sample, timestep, encoder_hidden_states):
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long, device=sample.device)
        t = t.reshape(-1).expand(sample.shape[0])
        t_emb = timestep_embedding(t, self.config.block_out_channels[0]).to(sample.dtype)
        temb = self.time_embedding(t_emb)
        x = self.conv_in(sample)
        skips = [x]
        for b in self.down_blocks:
            x, outs = b(x, temb, encoder_hidden_states)
            skips.extend(outs)
        x = self.mid_block(x, temb, encoder_hidden_states)
        for b in self.up_blocks:
            x = b(x, skips, temb, encoder_hidden_states)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return UNetOutput(x)


class AttnProcsLayers(nn.Module):
    """diffusers ``loaders.AttnProcsLayers``: ModuleList over the processor dict whose
    state_dict keys are remapped from ``layers.{i}`` to ``{attn name}.processor...``
    (2-export-checkpoint.py:630-634 saves exactly this state_dict)."""

    def __init__(self, procs: Dict[str, LoRAAttnProcessor]):
        super().__init__()
        self.layers = nn.ModuleList(procs.values())
        self.names = list(procs.keys())

    def state_dict(self, *a, **k):
        sd = super().state_dict(*a, **k)
        out = {}
        for key, v in sd.items():
            _, idx, rest = key.split(".", 2)
            out[f"{self.names[int(idx)]}.{rest}"] = v
        return out

    def load_named(self, sd):
        mine = {}
        for key, v in sd.items():
            for i, n in enumerate(self.names):
                if key.startswith(n + "."):
                    mine[f"layers.{i}.{key[len(n) + 1:]}"] = v
        return super().load_state_dict(mine, strict=True)


def make_unet_lora(unet: UNet2DConditionModel, rank: int, seed: int = 0) -> AttnProcsLayers:
    """Restates the injection loop at 1-main-debias.py:798-818."""
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    procs = {}
    boc = unet.config.block_out_channels
    for name in unet.attn_processors.keys():
        xdim = None if name.endswith("attn1.processor") else unet.config.cross_attention_dim
        if name.startswith("mid_block"):
            hidden = boc[-1]
        elif name.startswith("up_blocks"):
            hidden = list(reversed(boc))[int(name[len("up_blocks.")])]
        else:
            hidden = boc[int(name[len("down_blocks.")])]
        procs[name] = LoRAAttnProcessor(hidden, xdim, rank)
    unet.set_attn_processor(procs)
    torch.random.set_rng_state(g)
    return AttnProcsLayers(procs)
