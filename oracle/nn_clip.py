"""fp32 PyTorch restatement of transformers==4.30.0 ``CLIPTextModel`` with the
diffusers==0.19.3 text-encoder LoRA patch (``PatchedLoraProjection``).
TEST ORACLE -- PINNED against the installed transformers ``CLIPTextModel`` on shared random weights (causal + padding mask, quick_gelu;
tests/test_cpu.py::test_oracle_clip_text_pinned_against_transformers); the LoRA patch on top of it is restated (diffusers absent).

Reference call sites: exp-1-debias-gender/1-main-debias.py:726-729 (load),
:829-883 (``LoraLoaderMixin._modify_text_encoder(text_encoder, dtype=float32, rank,
patch_mlp=True)``), :1011-1014/:1078-1081 (forward with explicit attention_mask).
State-dict names follow transformers 4.30 (``text_model.`` prefix) and, once LoRA is
injected, the ``regular_linear_layer`` / ``lora_linear_layer`` split that
2-export-checkpoint.py:619-628 saves.
"""
from dataclasses import dataclass

import torch
import torch.nn as nn

from .nn_unet import LoRALinearLayer


@dataclass
class CLIPTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    layer_norm_eps: float = 1e-5


class PatchedLoraProjection(nn.Module):
    def __init__(self, regular: nn.Linear, rank: int, lora_scale: float = 1.0):
        super().__init__()
        self.regular_linear_layer = regular
        self.lora_linear_layer = LoRALinearLayer(regular.in_features, regular.out_features, rank)
        self.lora_scale = lora_scale

    def forward(self, x):
        return self.regular_linear_layer(x) + self.lora_scale * self.lora_linear_layer(x)


class CLIPAttention(nn.Module):
    def __init__(self, c: CLIPTextConfig):
        super().__init__()
        self.heads = c.num_attention_heads
        self.hd = c.hidden_size // self.heads
        self.k_proj = nn.Linear(c.hidden_size, c.hidden_size)
        self.v_proj = nn.Linear(c.hidden_size, c.hidden_size)
        self.q_proj = nn.Linear(c.hidden_size, c.hidden_size)
        self.out_proj = nn.Linear(c.hidden_size, c.hidden_size)

    def forward(self, x, bias):
        B, T, C = x.shape
        def split(t):
            return t.reshape(B, T, self.heads, self.hd).permute(0, 2, 1, 3)
        q = split(self.q_proj(x) * (self.hd ** -0.5))
        k, v = split(self.k_proj(x)), split(self.v_proj(x))
        w = torch.matmul(q, k.transpose(-1, -2)) + bias
        w = w.softmax(dim=-1)
        o = torch.matmul(w, v).permute(0, 2, 1, 3).reshape(B, T, C)
        return self.out_proj(o)


class CLIPMLP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.fc1 = nn.Linear(c.hidden_size, c.intermediate_size)
        self.fc2 = nn.Linear(c.intermediate_size, c.hidden_size)

    def forward(self, x):
        h = self.fc1(x)
        h = h * torch.sigmoid(1.702 * h)  # quick_gelu
        return self.fc2(h)


class CLIPEncoderLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self_attn = CLIPAttention(c)
        self.layer_norm1 = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.mlp = CLIPMLP(c)
        self.layer_norm2 = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)

    def forward(self, x, bias):
        x = x + self.self_attn(self.layer_norm1(x), bias)
        return x + self.mlp(self.layer_norm2(x))


class CLIPEmbeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.token_embedding = nn.Embedding(c.vocab_size, c.hidden_size)
        self.position_embedding = nn.Embedding(c.max_position_embeddings, c.hidden_size)


class CLIPEncoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layers = nn.ModuleList([CLIPEncoderLayer(c) for _ in range(c.num_hidden_layers)])


class CLIPTextTransformer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.embeddings = CLIPEmbeddings(c)
        self.encoder = CLIPEncoder(c)
        self.final_layer_norm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class CLIPTextModel(nn.Module):
    def __init__(self, cfg: CLIPTextConfig = CLIPTextConfig()):
        super().__init__()
        self.config = cfg
        self.text_model = CLIPTextTransformer(cfg)

    def forward(self, input_ids, attention_mask=None, fair=None):
        """``fair`` = (fair token ids [n], fair embedding table [n+1, D]): exp-2's ``FairEmbeddings.forward`` (gen-images.py:72-90) followed by
        ``text_model_forward`` (:190-271): positions whose id is a fair token get ``table[k+1] + position_embedding`` instead of the
        (resized) vocabulary embedding."""
        tm = self.text_model
        B, T = input_ids.shape
        pos = torch.arange(T, device=input_ids.device)
        V = tm.embeddings.token_embedding.weight.shape[0]
        x = tm.embeddings.token_embedding(input_ids.clamp(max=V - 1)) + tm.embeddings.position_embedding(pos)[None]
        if fair is not None:
            fair_ids, table = fair
            x = x.clone()
            for k, tid in enumerate(fair_ids.tolist()):
                sel = input_ids == tid
                x[sel] = (table[k + 1][None] + tm.embeddings.position_embedding(pos)[None].expand(B, T, -1)[sel]).to(x.dtype)
        neg = torch.finfo(x.dtype).min
        bias = torch.full((T, T), neg, dtype=x.dtype, device=x.device).triu(1)[None, None]
        if attention_mask is not None:
            pad = (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * neg  # _expand_mask
            bias = bias + pad
        for layer in tm.encoder.layers:
            x = layer(x, bias)
        x = tm.final_layer_norm(x)
        return (x,)


def modify_text_encoder(te: CLIPTextModel, rank: int, patch_mlp: bool = True, seed: int = 0):
    """``LoraLoaderMixin._modify_text_encoder`` (diffusers 0.19.3 loaders.py; called at 1-main-debias.py:831) restated: first the q/k/v/out_proj
    of EVERY layer's attention module, then -- with ``patch_mlp`` -- the fc1/fc2 of every layer's MLP (two separate loops, so the returned
    parameter LIST is attention-first; the reference's CustomModel / AdamW / EMAModel follow this list order, :836-842)."""
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    params = []
    for layer in te.text_model.encoder.layers:
        a = layer.self_attn
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            p = PatchedLoraProjection(getattr(a, n), rank)
            setattr(a, n, p)
            params.extend(p.lora_linear_layer.parameters())
    if patch_mlp:
        for layer in te.text_model.encoder.layers:
            for n in ("fc1", "fc2"):
                p = PatchedLoraProjection(getattr(layer.mlp, n), rank)
                setattr(layer.mlp, n, p)
                params.extend(p.lora_linear_layer.parameters())
    torch.random.set_rng_state(g)
    return params
