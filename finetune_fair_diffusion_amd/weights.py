"""Parameter inventories (diffusers / transformers / torchvision state-dict names and shapes) for
the four frozen networks on the path, and a deterministic synthetic initialiser for them.

The names are the drop-in contract (SURVEY.md 8b "Checkpoint format"): SD-v1.5 weights in the
diffusers layout and the reference's exported LoRA files load by key.  There is no network in the
build/bench environment, so benchmarks and tests use ``synthetic_state_dict`` (variance-preserving
random weights) -- "data": "synthetic" in bench.py.
"""
from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple

import torch


@dataclass
class UNetConfig:  # diffusers UNet2DConditionModel config keys used by the reference (:800-809)
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    attention_head_dim: int = 8  # number of heads in SD-v1.5
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D")
    up_block_types: Tuple[str, ...] = ("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D")
    sample_size: int = 64


@dataclass
class VAEConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215


@dataclass
class CLIPTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    layer_norm_eps: float = 1e-5


# ------------------------------------------------------------------------------------------ U-Net
def _resnet(sd, p, cin, cout, temb):
    sd[p + "norm1.weight"] = (cin,); sd[p + "norm1.bias"] = (cin,)
    sd[p + "conv1.weight"] = (cout, cin, 3, 3); sd[p + "conv1.bias"] = (cout,)
    if temb:
        sd[p + "time_emb_proj.weight"] = (cout, temb); sd[p + "time_emb_proj.bias"] = (cout,)
    sd[p + "norm2.weight"] = (cout,); sd[p + "norm2.bias"] = (cout,)
    sd[p + "conv2.weight"] = (cout, cout, 3, 3); sd[p + "conv2.bias"] = (cout,)
    if cin != cout:
        sd[p + "conv_shortcut.weight"] = (cout, cin, 1, 1); sd[p + "conv_shortcut.bias"] = (cout,)


def _transformer(sd, p, c, xdim):
    sd[p + "norm.weight"] = (c,); sd[p + "norm.bias"] = (c,)
    sd[p + "proj_in.weight"] = (c, c, 1, 1); sd[p + "proj_in.bias"] = (c,)
    b = p + "transformer_blocks.0."
    sd[b + "norm1.weight"] = (c,); sd[b + "norm1.bias"] = (c,)
    for a, kv in (("attn1", c), ("attn2", xdim)):
        if a == "attn2":
            sd[b + "norm2.weight"] = (c,); sd[b + "norm2.bias"] = (c,)
        sd[b + a + ".to_q.weight"] = (c, c)
        sd[b + a + ".to_k.weight"] = (c, kv)
        sd[b + a + ".to_v.weight"] = (c, kv)
        sd[b + a + ".to_out.0.weight"] = (c, c); sd[b + a + ".to_out.0.bias"] = (c,)
    sd[b + "norm3.weight"] = (c,); sd[b + "norm3.bias"] = (c,)
    sd[b + "ff.net.0.proj.weight"] = (8 * c, c); sd[b + "ff.net.0.proj.bias"] = (8 * c,)
    sd[b + "ff.net.2.weight"] = (c, 4 * c); sd[b + "ff.net.2.bias"] = (c,)
    sd[p + "proj_out.weight"] = (c, c, 1, 1); sd[p + "proj_out.bias"] = (c,)


def unet_param_shapes(cfg: UNetConfig) -> "OrderedDict[str, tuple]":
    sd = OrderedDict()
    boc, xdim, n = cfg.block_out_channels, cfg.cross_attention_dim, cfg.layers_per_block
    temb = boc[0] * 4
    sd["conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3); sd["conv_in.bias"] = (boc[0],)
    sd["time_embedding.linear_1.weight"] = (temb, boc[0]); sd["time_embedding.linear_1.bias"] = (temb,)
    sd["time_embedding.linear_2.weight"] = (temb, temb); sd["time_embedding.linear_2.bias"] = (temb,)
    cout = boc[0]
    for i, t in enumerate(cfg.down_block_types):
        cin, cout = cout, boc[i]
        for j in range(n):
            _resnet(sd, f"down_blocks.{i}.resnets.{j}.", cin if j == 0 else cout, cout, temb)
        if t.startswith("CrossAttn"):
            for j in range(n):
                _transformer(sd, f"down_blocks.{i}.attentions.{j}.", cout, xdim)
        if i != len(boc) - 1:
            sd[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            sd[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (cout,)
    c = boc[-1]
    _resnet(sd, "mid_block.resnets.0.", c, c, temb)
    _transformer(sd, "mid_block.attentions.0.", c, xdim)
    _resnet(sd, "mid_block.resnets.1.", c, c, temb)
    rev = list(reversed(boc))
    cout = rev[0]
    for i, t in enumerate(cfg.up_block_types):
        cprev, cout = cout, rev[i]
        cin = rev[min(i + 1, len(boc) - 1)]
        for j in range(n + 1):
            skip = cin if j == n else cout
            rin = cprev if j == 0 else cout
            _resnet(sd, f"up_blocks.{i}.resnets.{j}.", rin + skip, cout, temb)
        if t.startswith("CrossAttn"):
            for j in range(n + 1):
                _transformer(sd, f"up_blocks.{i}.attentions.{j}.", cout, xdim)
        if i != len(boc) - 1:
            sd[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            sd[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
    sd["conv_norm_out.weight"] = (boc[0],); sd["conv_norm_out.bias"] = (boc[0],)
    sd["conv_out.weight"] = (cfg.out_channels, boc[0], 3, 3); sd["conv_out.bias"] = (cfg.out_channels,)
    return sd


def unet_attn_names(cfg: UNetConfig):
    """Attention-processor keys in diffusers ``unet.attn_processors`` order (down, up, mid -- see SURVEY 8b)."""
    names = []
    n = cfg.layers_per_block
    for i, t in enumerate(cfg.down_block_types):
        if t.startswith("CrossAttn"):
            for j in range(n):
                for a in ("attn1", "attn2"):
                    names.append(f"down_blocks.{i}.attentions.{j}.transformer_blocks.0.{a}.processor")
    for i, t in enumerate(cfg.up_block_types):
        if t.startswith("CrossAttn"):
            for j in range(n + 1):
                for a in ("attn1", "attn2"):
                    names.append(f"up_blocks.{i}.attentions.{j}.transformer_blocks.0.{a}.processor")
    for a in ("attn1", "attn2"):
        names.append(f"mid_block.attentions.0.transformer_blocks.0.{a}.processor")
    return names


def unet_lora_param_shapes(cfg: UNetConfig, rank: int) -> "OrderedDict[str, tuple]":
    """``AttnProcsLayers(unet.attn_processors).state_dict()`` keys (2-export-checkpoint.py:630-634)."""
    sd = OrderedDict()
    boc = cfg.block_out_channels
    for name in unet_attn_names(cfg):
        if name.startswith("mid_block"):
            hidden = boc[-1]
        elif name.startswith("up_blocks"):
            hidden = list(reversed(boc))[int(name[len("up_blocks.")])]
        else:
            hidden = boc[int(name[len("down_blocks.")])]
        kv = hidden if name.endswith("attn1.processor") else cfg.cross_attention_dim
        for proj, cin in (("to_q", hidden), ("to_k", kv), ("to_v", kv), ("to_out", hidden)):
            sd[f"{name}.{proj}_lora.down.weight"] = (rank, cin)
            sd[f"{name}.{proj}_lora.up.weight"] = (hidden, rank)
    return sd


# ------------------------------------------------------------------------------------------ VAE decoder
def vae_param_shapes(cfg: VAEConfig) -> "OrderedDict[str, tuple]":
    sd = OrderedDict()
    boc, n = cfg.block_out_channels, cfg.layers_per_block
    L = cfg.latent_channels
    sd["post_quant_conv.weight"] = (L, L, 1, 1); sd["post_quant_conv.bias"] = (L,)
    c = boc[-1]
    sd["decoder.conv_in.weight"] = (c, L, 3, 3); sd["decoder.conv_in.bias"] = (c,)
    _resnet(sd, "decoder.mid_block.resnets.0.", c, c, 0)
    a = "decoder.mid_block.attentions.0."
    sd[a + "group_norm.weight"] = (c,); sd[a + "group_norm.bias"] = (c,)
    for nme in ("to_q", "to_k", "to_v", "to_out.0"):
        sd[a + nme + ".weight"] = (c, c); sd[a + nme + ".bias"] = (c,)
    _resnet(sd, "decoder.mid_block.resnets.1.", c, c, 0)
    rev = list(reversed(boc))
    cout = rev[0]
    for i in range(len(rev)):
        cin, cout = cout, rev[i]
        for j in range(n + 1):
            _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}.", cin if j == 0 else cout, cout, 0)
        if i != len(rev) - 1:
            sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
    sd["decoder.conv_norm_out.weight"] = (boc[0],); sd["decoder.conv_norm_out.bias"] = (boc[0],)
    sd["decoder.conv_out.weight"] = (cfg.out_channels, boc[0], 3, 3); sd["decoder.conv_out.bias"] = (cfg.out_channels,)
    return sd


# ------------------------------------------------------------------------------------------ CLIP text encoder
TE_LORA_TARGETS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj", "mlp.fc1", "mlp.fc2")


def clip_param_shapes(cfg: CLIPTextConfig) -> "OrderedDict[str, tuple]":
    sd = OrderedDict()
    D, I = cfg.hidden_size, cfg.intermediate_size
    sd["text_model.embeddings.token_embedding.weight"] = (cfg.vocab_size, D)
    sd["text_model.embeddings.position_embedding.weight"] = (cfg.max_position_embeddings, D)
    for i in range(cfg.num_hidden_layers):
        p = f"text_model.encoder.layers.{i}."
        for nme in ("k_proj", "v_proj", "q_proj", "out_proj"):
            sd[p + f"self_attn.{nme}.weight"] = (D, D); sd[p + f"self_attn.{nme}.bias"] = (D,)
        sd[p + "layer_norm1.weight"] = (D,); sd[p + "layer_norm1.bias"] = (D,)
        sd[p + "mlp.fc1.weight"] = (I, D); sd[p + "mlp.fc1.bias"] = (I,)
        sd[p + "mlp.fc2.weight"] = (D, I); sd[p + "mlp.fc2.bias"] = (D,)
        sd[p + "layer_norm2.weight"] = (D,); sd[p + "layer_norm2.bias"] = (D,)
    sd["text_model.final_layer_norm.weight"] = (D,); sd["text_model.final_layer_norm.bias"] = (D,)
    return sd


def clip_lora_param_shapes(cfg: CLIPTextConfig, rank: int) -> "OrderedDict[str, tuple]":
    """LoRA tensors in ``_modify_text_encoder(..., patch_mlp=True)`` registration order (:829-844)."""
    sd = OrderedDict()
    D, I = cfg.hidden_size, cfg.intermediate_size
    for i in range(cfg.num_hidden_layers):
        for tgt in TE_LORA_TARGETS:
            cin, cout = (I, D) if tgt == "mlp.fc2" else ((D, I) if tgt == "mlp.fc1" else (D, D))
            p = f"text_model.encoder.layers.{i}.{tgt}.lora_linear_layer."
            sd[p + "down.weight"] = (rank, cin)
            sd[p + "up.weight"] = (cout, rank)
    return sd


# ------------------------------------------------------------------------------------------ MobileNetV3-Large
MBV3_SETTINGS = [  # kernel, expanded, out, use_se, activation, stride
    (3, 16, 16, False, "relu", 1), (3, 64, 24, False, "relu", 2), (3, 72, 24, False, "relu", 1),
    (5, 72, 40, True, "relu", 2), (5, 120, 40, True, "relu", 1), (5, 120, 40, True, "relu", 1),
    (3, 240, 80, False, "hardswish", 2), (3, 200, 80, False, "hardswish", 1), (3, 184, 80, False, "hardswish", 1),
    (3, 184, 80, False, "hardswish", 1), (3, 480, 112, True, "hardswish", 1), (3, 672, 112, True, "hardswish", 1),
    (5, 672, 160, True, "hardswish", 2), (5, 960, 160, True, "hardswish", 1), (5, 960, 160, True, "hardswish", 1),
]


def make_divisible(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def _cba(sd, p, cin, cout, k, groups=1):
    sd[p + "0.weight"] = (cout, cin // groups, k, k)
    for n in ("weight", "bias", "running_mean", "running_var"):
        sd[p + "1." + n] = (cout,)
    sd[p + "1.num_batches_tracked"] = ()


def mobilenet_param_shapes(num_classes=80) -> "OrderedDict[str, tuple]":
    sd = OrderedDict()
    _cba(sd, "features.0.", 3, 16, 3)
    cin = 16
    for i, (k, exp, cout, se, act, s) in enumerate(MBV3_SETTINGS):
        p = f"features.{i + 1}.block."
        j = 0
        if exp != cin:
            _cba(sd, p + f"{j}.", cin, exp, 1); j += 1
        _cba(sd, p + f"{j}.", exp, exp, k, groups=exp); j += 1
        if se:
            sq = make_divisible(exp // 4, 8)
            sd[p + f"{j}.fc1.weight"] = (sq, exp, 1, 1); sd[p + f"{j}.fc1.bias"] = (sq,)
            sd[p + f"{j}.fc2.weight"] = (exp, sq, 1, 1); sd[p + f"{j}.fc2.bias"] = (exp,)
            j += 1
        _cba(sd, p + f"{j}.", exp, cout, 1)
        cin = cout
    _cba(sd, "features.16.", cin, 960, 1)
    sd["classifier.0.weight"] = (1280, 960); sd["classifier.0.bias"] = (1280,)
    sd["classifier.3.weight"] = (num_classes, 1280); sd["classifier.3.bias"] = (num_classes,)
    return sd


# ------------------------------------------------------------------------------------------ image encoders of the regularisers
@dataclass
class ViTConfig:
    """``kind`` "clip": transformers CLIPVisionModelWithProjection (laion/CLIP-ViT-H-14, :948-957);
    "dino": facebookresearch/dinov2 ``dinov2_vitb14`` (:960-962)."""
    kind: str = "clip"
    image_size: int = 224
    patch_size: int = 14
    hidden_size: int = 1280
    num_hidden_layers: int = 32
    num_attention_heads: int = 16
    intermediate_size: int = 5120
    projection_dim: int = 1024
    layer_norm_eps: float = 1e-5
    pos_grid: int = 16


CLIP_VIT_H14 = ViTConfig()
DINOV2_VITB14 = ViTConfig(kind="dino", hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                          projection_dim=0, layer_norm_eps=1e-6, pos_grid=37)
CLIP_IMAGE_MEAN, CLIP_IMAGE_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
DINO_IMAGE_MEAN, DINO_IMAGE_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def vit_param_shapes(c: ViTConfig) -> "OrderedDict[str, tuple]":
    sd = OrderedDict()
    D, I, P = c.hidden_size, c.intermediate_size, c.patch_size
    if c.kind == "clip":
        e = "vision_model.embeddings."
        sd[e + "class_embedding"] = (D,)
        sd[e + "patch_embedding.weight"] = (D, 3, P, P)
        sd[e + "position_embedding.weight"] = ((c.image_size // P) ** 2 + 1, D)
        sd["vision_model.pre_layrnorm.weight"] = (D,); sd["vision_model.pre_layrnorm.bias"] = (D,)
        for i in range(c.num_hidden_layers):
            p = f"vision_model.encoder.layers.{i}."
            for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
                sd[p + f"self_attn.{n}.weight"] = (D, D); sd[p + f"self_attn.{n}.bias"] = (D,)
            sd[p + "layer_norm1.weight"] = (D,); sd[p + "layer_norm1.bias"] = (D,)
            sd[p + "mlp.fc1.weight"] = (I, D); sd[p + "mlp.fc1.bias"] = (I,)
            sd[p + "mlp.fc2.weight"] = (D, I); sd[p + "mlp.fc2.bias"] = (D,)
            sd[p + "layer_norm2.weight"] = (D,); sd[p + "layer_norm2.bias"] = (D,)
        sd["vision_model.post_layernorm.weight"] = (D,); sd["vision_model.post_layernorm.bias"] = (D,)
        sd["visual_projection.weight"] = (c.projection_dim, D)
    else:
        sd["cls_token"] = (1, 1, D)
        sd["pos_embed"] = (1, 1 + c.pos_grid ** 2, D)
        sd["patch_embed.proj.weight"] = (D, 3, P, P); sd["patch_embed.proj.bias"] = (D,)
        for i in range(c.num_hidden_layers):
            p = f"blocks.{i}."
            sd[p + "norm1.weight"] = (D,); sd[p + "norm1.bias"] = (D,)
            sd[p + "attn.qkv.weight"] = (3 * D, D); sd[p + "attn.qkv.bias"] = (3 * D,)
            sd[p + "attn.proj.weight"] = (D, D); sd[p + "attn.proj.bias"] = (D,)
            sd[p + "ls1.gamma"] = (D,)
            sd[p + "norm2.weight"] = (D,); sd[p + "norm2.bias"] = (D,)
            sd[p + "mlp.fc1.weight"] = (I, D); sd[p + "mlp.fc1.bias"] = (I,)
            sd[p + "mlp.fc2.weight"] = (D, I); sd[p + "mlp.fc2.bias"] = (D,)
            sd[p + "ls2.gamma"] = (D,)
        sd["norm.weight"] = (D,); sd["norm.bias"] = (D,)
    return sd


# ------------------------------------------------------------------------------------------ face-feature network (face-realism term)
SFNET20_LAYERS = (1, 2, 4, 1)
SFNET20_CHANNELS = (64, 128, 256, 512)


def sfnet20_param_shapes(channels=SFNET20_CHANNELS, out_channel=512, in_size=112) -> "OrderedDict[str, tuple]":
    """opensphere ``sfnet20`` without norm layers (opensphere/model/backbone/sfnet.py:123-202, 252-261): per stage a stride-2 ConvBlock
    (``layerK.0.conv1``) and n BasicBlocks (``layerK.j.conv1/conv2``), then ``fc`` on the NCHW-flattened 7x7 map."""
    sd = OrderedDict()
    cin = 3
    for i, (c, n) in enumerate(zip(channels, SFNET20_LAYERS)):
        sd[f"layer{i + 1}.0.conv1.weight"] = (c, cin, 3, 3); sd[f"layer{i + 1}.0.conv1.bias"] = (c,)
        for j in range(1, n + 1):
            for k in ("conv1", "conv2"):
                sd[f"layer{i + 1}.{j}.{k}.weight"] = (c, c, 3, 3); sd[f"layer{i + 1}.{j}.{k}.bias"] = (c,)
        cin = c
    sd["fc.weight"] = (out_channel, channels[3] * (in_size // 16) ** 2); sd["fc.bias"] = (out_channel,)
    return sd


# ------------------------------------------------------------------------------------------ synthetic init
def synthetic_state_dict(shapes, seed=0, device="cpu", gain=1.0, dtype=torch.float32):
    """Variance-preserving random weights: W ~ N(0, gain^2/fan_in), biases ~ N(0, 0.02^2),
    norm scales 1 + 0.1 N, norm shifts 0.05 N, BatchNorm running stats non-trivial."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = OrderedDict()
    for name, shape in shapes.items():
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t = torch.zeros((), dtype=torch.long)
            sd[name] = t.to(device)
            continue
        if leaf == "running_mean":
            t = torch.randn(shape, generator=g) * 0.1
        elif leaf == "running_var":
            t = torch.rand(shape, generator=g) * 0.5 + 0.75
        elif len(shape) == 1:
            is_norm = any(k in name for k in ("norm", ".1.weight", ".1.bias", "layer_norm"))
            if leaf == "class_embedding":
                t = torch.randn(shape, generator=g) * 0.5
            elif leaf == "gamma":                                  # DINOv2 LayerScale
                t = 0.5 + 0.1 * torch.randn(shape, generator=g)
            elif leaf == "weight":
                t = 1.0 + 0.1 * torch.randn(shape, generator=g)
            else:
                t = (0.05 if is_norm else 0.02) * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            if leaf in ("cls_token", "pos_embed") or "position_embedding" in name or "token_embedding" in name:
                t = torch.randn(shape, generator=g) * 0.5
            elif "lora" in name and ".up." in name:
                t = torch.zeros(shape)
            elif "lora" in name and ".down." in name:
                t = torch.randn(shape, generator=g) / shape[0]
            else:
                t = torch.randn(shape, generator=g) * (gain / fan_in ** 0.5)
        sd[name] = t.to(dtype).to(device)
    return sd
