"""fp32 PyTorch restatement of the two image encoders behind the reference's image-semantics regularisers
(exp-1-debias-gender/1-main-debias.py:948-966 load, :1139-1175 ``get_clip_feat`` / ``get_dino_feat``,
:1860-1862 / :1905-1910 use).  TEST ORACLE.

* ``CLIPVisionModelWithProjection`` -- transformers==4.30.0 ``models/clip/modeling_clip.py`` (laion/CLIP-ViT-H-14:
  width 1280, 32 layers, 16 heads, MLP 5120, patch 14, 224 px, exact GELU, LN eps 1e-5, projection 1024 w/o bias).
  State-dict names are transformers' (``vision_model.embeddings.class_embedding`` ... ``visual_projection.weight``).
  PINNED against the installed transformers implementation on random weights (tests/test_cpu.py).
* ``DinoVisionTransformer`` -- facebookresearch/dinov2 ``models/vision_transformer.py`` via torch.hub (``dinov2_vitb14``:
  width 768, 12 layers, 12 heads, MLP 3072, patch 14, LayerScale, LN eps 1e-6, position table trained on 37x37 patches and
  bicubically interpolated with the ``+0.1`` scale-factor offset, output = final-norm CLS token).  The hub repository is
  neither vendored nor installed and cannot be fetched; PINNED (round 4) against the installed transformers ``Dinov2Model`` -- an independent port
  of the same network -- on random weights, with and without position-table interpolation (tests/test_cpu.py, 1e-5).
"""
import math
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class ViTConfig:
    kind: str = "clip"              # "clip" | "dino"
    image_size: int = 224
    patch_size: int = 14
    hidden_size: int = 1280
    num_hidden_layers: int = 32
    num_attention_heads: int = 16
    intermediate_size: int = 5120
    projection_dim: int = 1024      # clip only
    layer_norm_eps: float = 1e-5
    pos_grid: int = 16              # rows/cols of the stored position table (dino: 37, interpolated to image_size/patch_size)


CLIP_VIT_H14 = ViTConfig()
DINOV2_VITB14 = ViTConfig(kind="dino", hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                          projection_dim=0, layer_norm_eps=1e-6, pos_grid=37)
CLIP_IMAGE_MEAN, CLIP_IMAGE_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)   # CLIPImageProcessor defaults
DINO_IMAGE_MEAN, DINO_IMAGE_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)                                   # :963-964


def _mha(x, q, k, v, heads):
    B, T, D = x.shape
    d = D // heads
    sh = lambda t: t.view(B, T, heads, d).transpose(1, 2)  # noqa: E731
    p = torch.softmax((sh(q) * d ** -0.5) @ sh(k).transpose(-1, -2), dim=-1)
    return (p @ sh(v)).transpose(1, 2).reshape(B, T, D)


class CLIPEncoderLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        D = c.hidden_size
        self.self_attn = nn.Module()
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            setattr(self.self_attn, n, nn.Linear(D, D))
        self.layer_norm1 = nn.LayerNorm(D, eps=c.layer_norm_eps)
        self.mlp = nn.Module()
        self.mlp.fc1, self.mlp.fc2 = nn.Linear(D, c.intermediate_size), nn.Linear(c.intermediate_size, D)
        self.layer_norm2 = nn.LayerNorm(D, eps=c.layer_norm_eps)
        self.heads = c.num_attention_heads

    def forward(self, x):
        h = self.layer_norm1(x)
        a = self.self_attn
        x = x + a.out_proj(_mha(h, a.q_proj(h), a.k_proj(h), a.v_proj(h), self.heads))
        h = self.layer_norm2(x)
        return x + self.mlp.fc2(F.gelu(self.mlp.fc1(h)))


class CLIPVisionModelWithProjection(nn.Module):
    def __init__(self, c: ViTConfig):
        super().__init__()
        D, n = c.hidden_size, (c.image_size // c.patch_size) ** 2
        vm = self.vision_model = nn.Module()
        vm.embeddings = nn.Module()
        vm.embeddings.class_embedding = nn.Parameter(torch.randn(D))
        vm.embeddings.patch_embedding = nn.Conv2d(3, D, c.patch_size, c.patch_size, bias=False)
        vm.embeddings.position_embedding = nn.Embedding(n + 1, D)
        vm.pre_layrnorm = nn.LayerNorm(D, eps=c.layer_norm_eps)        # (sic) transformers' attribute name
        vm.encoder = nn.Module()
        vm.encoder.layers = nn.ModuleList([CLIPEncoderLayer(c) for _ in range(c.num_hidden_layers)])
        vm.post_layernorm = nn.LayerNorm(D, eps=c.layer_norm_eps)
        self.visual_projection = nn.Linear(D, c.projection_dim, bias=False)

    def forward(self, pixel_values):
        vm = self.vision_model
        e = vm.embeddings
        p = e.patch_embedding(pixel_values).flatten(2).transpose(1, 2)
        x = torch.cat([e.class_embedding.expand(p.shape[0], 1, -1), p], dim=1) + e.position_embedding.weight[None]
        x = vm.pre_layrnorm(x)
        for l in vm.encoder.layers:
            x = l(x)
        return self.visual_projection(vm.post_layernorm(x[:, 0]))       # .image_embeds


class DinoBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        D = c.hidden_size
        self.norm1 = nn.LayerNorm(D, eps=c.layer_norm_eps)
        self.attn = nn.Module()
        self.attn.qkv, self.attn.proj = nn.Linear(D, 3 * D), nn.Linear(D, D)
        self.ls1 = nn.Module()
        self.ls1.gamma = nn.Parameter(torch.ones(D))
        self.norm2 = nn.LayerNorm(D, eps=c.layer_norm_eps)
        self.mlp = nn.Module()
        self.mlp.fc1, self.mlp.fc2 = nn.Linear(D, c.intermediate_size), nn.Linear(c.intermediate_size, D)
        self.ls2 = nn.Module()
        self.ls2.gamma = nn.Parameter(torch.ones(D))
        self.heads = c.num_attention_heads

    def forward(self, x):
        h = self.norm1(x)
        q, k, v = self.attn.qkv(h).chunk(3, dim=-1)
        x = x + self.ls1.gamma * self.attn.proj(_mha(h, q, k, v, self.heads))
        h = self.norm2(x)
        return x + self.ls2.gamma * self.mlp.fc2(F.gelu(self.mlp.fc1(h)))


def interpolate_pos_encoding(pos_embed, grid_out):
    """dinov2 ``interpolate_pos_encoding``: [1, 1+M*M, D] -> [1, 1+g*g, D]; identity when the grids agree."""
    N = pos_embed.shape[1] - 1
    M = int(math.sqrt(N))
    if M == grid_out:
        return pos_embed
    D = pos_embed.shape[-1]
    w0 = h0 = grid_out + 0.1
    patch = F.interpolate(pos_embed[:, 1:].float().reshape(1, M, M, D).permute(0, 3, 1, 2), scale_factor=(w0 / M, h0 / M), mode="bicubic")
    assert patch.shape[-1] == grid_out and patch.shape[-2] == grid_out
    return torch.cat([pos_embed[:, :1].float(), patch.permute(0, 2, 3, 1).reshape(1, -1, D)], dim=1).to(pos_embed.dtype)


class DinoVisionTransformer(nn.Module):
    def __init__(self, c: ViTConfig):
        super().__init__()
        D = c.hidden_size
        self.cfg = c
        self.cls_token = nn.Parameter(torch.randn(1, 1, D) * 0.02)
        self.pos_embed = nn.Parameter(torch.randn(1, 1 + c.pos_grid ** 2, D) * 0.02)
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv2d(3, D, c.patch_size, c.patch_size)
        self.blocks = nn.ModuleList([DinoBlock(c) for _ in range(c.num_hidden_layers)])
        self.norm = nn.LayerNorm(D, eps=c.layer_norm_eps)

    def forward(self, x):
        p = self.patch_embed.proj(x).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls_token.expand(p.shape[0], -1, -1), p], dim=1)
        x = x + interpolate_pos_encoding(self.pos_embed, self.cfg.image_size // self.cfg.patch_size)
        for b in self.blocks:
            x = b(x)
        return self.norm(x)[:, 0]                                        # head = Identity on x_norm_clstoken


def build(cfg: ViTConfig, state_dict=None):
    m = (CLIPVisionModelWithProjection if cfg.kind == "clip" else DinoVisionTransformer)(cfg)
    if state_dict is not None:
        m.load_state_dict({k: v.float() for k, v in state_dict.items()}, strict=True)
    return m.float().eval().requires_grad_(False)


def image_features(model, images, mean, std, normalize=True):
    """get_clip_feat / get_dino_feat (:1139-1175): images in [-1,1] -> ((x+1)/2 - mean)/std -> encoder -> L2-normalised fp32."""
    m = torch.tensor(mean, dtype=images.dtype).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=images.dtype).view(1, 3, 1, 1)
    e = model(((images + 1) * 0.5 - m) / s).float()
    return F.normalize(e, dim=-1) if normalize else e
