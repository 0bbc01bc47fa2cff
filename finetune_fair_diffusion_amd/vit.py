"""The two frozen image encoders of the image-semantics regularisers on MI355X: CLIP ViT-H/14 (transformers
``CLIPVisionModelWithProjection``) and DINOv2 ViT-B/14 (``torch.hub`` ``dinov2_vitb14``) -- reference load sites
exp-1-debias-gender/1-main-debias.py:948-964, use ``get_clip_feat`` / ``get_dino_feat`` :1139-1175 on
``Resize(224)(images)`` (:1860-1862 for the frozen model's images, :1905-1910 with gradient).

Both are pre-LN ViTs and share one engine: patches are unfolded by ``fd_patchify`` (with the per-channel
normalisation fused) so the patch embedding is a GEMM; tokens are padded from 257 to 264 rows per image so that the
MFMA attention kernels see 8-aligned rows (pad keys are masked by ``Tk``; pad queries carry zero gradient);
q-scale, LayerScale and the exact-GELU MLP use the same GEMM / LayerNorm / flash-attention kernels as the U-Net.
Weights are frozen: the backward produces only the gradient w.r.t. the 224x224 input, explicitly (no autograd).
"""
import math

import torch
import torch.nn.functional as F

from . import ops
from .layers import F16, F32, Linear, Norm
from .weights import ViTConfig


def _interpolate_pos(pos_embed, grid_out):
    """dinov2 ``interpolate_pos_encoding`` (bicubic, ``+0.1`` scale-factor offset): one-off weight preparation on the host."""
    N = pos_embed.shape[1] - 1
    M = int(math.sqrt(N))
    if M == grid_out:
        return pos_embed.float()
    D = pos_embed.shape[-1]
    w0 = grid_out + 0.1
    patch = F.interpolate(pos_embed[:, 1:].float().reshape(1, M, M, D).permute(0, 3, 1, 2), scale_factor=(w0 / M, w0 / M), mode="bicubic")
    if patch.shape[-1] != grid_out:
        raise ValueError(f"position table {M}x{M} does not interpolate to {grid_out}x{grid_out}")
    return torch.cat([pos_embed[:, :1].float(), patch.permute(0, 2, 3, 1).reshape(1, -1, D)], dim=1)


class _Lin:
    """Frozen y = x W^T + b with an optional per-output scale folded in (DINOv2 LayerScale)."""

    def __init__(self, w, b, dev, scale=None):
        w, b = w.float(), (b.float() if b is not None else None)
        if scale is not None:
            w, b = w * scale.float()[:, None], b * scale.float()
        self.w = w.to(dev, F16).contiguous()
        self.bias = b.to(dev, F32).contiguous() if b is not None else None
        self.wT = self.w.t().contiguous()


class VisionTransformer:
    def __init__(self, cfg: ViTConfig, sd, device, mean, std):
        self.config, self.device = cfg, device
        self.mean, self.std = tuple(mean), tuple(std)
        D, P = cfg.hidden_size, cfg.patch_size
        self.g = cfg.image_size // P
        self.T = self.g * self.g + 1
        self.Tp = (self.T + 7) // 8 * 8
        self.Kp = (3 * P * P + 7) // 8 * 8
        self.H, self.d = cfg.num_attention_heads, D // cfg.num_attention_heads
        dev = device
        clip = cfg.kind == "clip"
        pe = "vision_model.embeddings.patch_embedding" if clip else "patch_embed.proj"
        w = torch.zeros(D, self.Kp)
        w[:, :3 * P * P] = sd[pe + ".weight"].float().reshape(D, -1)
        self.patch = _Lin(w, sd.get(pe + ".bias"), dev)
        if clip:
            cls = sd["vision_model.embeddings.class_embedding"].float().view(1, D)
            pos = sd["vision_model.embeddings.position_embedding.weight"].float()
        else:
            cls = sd["cls_token"].float().view(1, D)
            pos = _interpolate_pos(sd["pos_embed"], self.g)[0]
        self.cls_pos0 = (cls[0] + pos[0]).to(dev, F16)
        self.pos_patch = pos[1:].to(dev, F16).contiguous()                  # [g*g, D], added as the GEMM residual
        self.pre_ln = Norm(sd, "vision_model.pre_layrnorm", dev) if clip else None
        self.layers = []
        for i in range(cfg.num_hidden_layers):
            if clip:
                p = f"vision_model.encoder.layers.{i}."
                q, k, v = (_Lin(sd[p + f"self_attn.{n}_proj.weight"], sd[p + f"self_attn.{n}_proj.bias"], dev) for n in "qkv")
                o = _Lin(sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], dev)
                L = dict(ln1=Norm(sd, p + "layer_norm1", dev), ln2=Norm(sd, p + "layer_norm2", dev), q=q, k=k, v=v, o=o,
                         fc1=_Lin(sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], dev), fc2=_Lin(sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"], dev))
            else:
                p = f"blocks.{i}."
                W, b = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
                q, k, v = (_Lin(W[j * D:(j + 1) * D], b[j * D:(j + 1) * D], dev) for j in range(3))
                L = dict(ln1=Norm(sd, p + "norm1", dev), ln2=Norm(sd, p + "norm2", dev), q=q, k=k, v=v,
                         o=_Lin(sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], dev, scale=sd[p + "ls1.gamma"]),
                         fc1=_Lin(sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], dev),
                         fc2=_Lin(sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"], dev, scale=sd[p + "ls2.gamma"]))
            self.layers.append(L)
        self.final_ln = Norm(sd, "vision_model.post_layernorm" if clip else "norm", dev)
        self.proj = _Lin(sd["visual_projection.weight"], None, dev) if clip else None
        self.out_dim = cfg.projection_dim if clip else D
        self._ctx = None

    # ------------------------------------------------------------------ forward
    def forward(self, chips, record=False):
        """chips [N,3,S,S] fp16 NCHW in [-1,1] (already resized).  Returns the raw embedding [N, out_dim] fp32."""
        cfg = self.config
        N = chips.shape[0]
        D, H, d, T, Tp, eps = cfg.hidden_size, self.H, self.d, self.T, self.Tp, cfg.layer_norm_eps
        patches = ops.patchify(chips, self.mean, self.std, cfg.patch_size, self.Kp)            # [N*g*g, Kp]
        x = torch.zeros((N, Tp, D), dtype=F16, device=self.device)
        x[:, 0] = self.cls_pos0
        ops.gemm_batched_into(patches, self.patch.w, x[:, 1:T], self.patch.bias, self.pos_patch, N, self.g * self.g)
        x = x.view(N * Tp, D)
        ctx = dict(layers=[], N=N) if record else None
        if self.pre_ln is not None:
            x0 = x
            x, s0 = ops.layernorm(x0, self.pre_ln.gamma, self.pre_ln.beta, eps, save_stats=True)
            if record:
                ctx.update(x0=x0, s0=s0)
        for L in self.layers:
            n1, s1 = ops.layernorm(x, L["ln1"].gamma, L["ln1"].beta, eps, save_stats=True)
            q = ops.gemm(n1, L["q"].w, bias=L["q"].bias)
            k = ops.gemm(n1, L["k"].w, bias=L["k"].bias)
            v = ops.gemm(n1, L["v"].w, bias=L["v"].bias)
            a, lse = ops.attn_fwd(q, k, v, N, H, Tp, T, d, need_lse=True, kv_rows=Tp)
            h1 = ops.gemm(a, L["o"].w, bias=L["o"].bias, residual=x)
            n2, s2 = ops.layernorm(h1, L["ln2"].gamma, L["ln2"].beta, eps, save_stats=True)
            if record:
                z = ops.gemm(n2, L["fc1"].w, bias=L["fc1"].bias)
                m = ops.act_fwd(z, "gelu")
            else:
                z, m = None, ops.gemm(n2, L["fc1"].w, bias=L["fc1"].bias, act="gelu")
            h2 = ops.gemm(m, L["fc2"].w, bias=L["fc2"].bias, residual=h1)
            if record:
                ctx["layers"].append(dict(x=x, s1=s1, q=q, k=k, v=v, a=a, lse=lse, h1=h1, s2=s2, z=z))
            x = h2
        cls = x.view(N, Tp, D)[:, 0].contiguous()
        y, sf = ops.layernorm(cls, self.final_ln.gamma, self.final_ln.beta, eps, save_stats=True)
        e = ops.gemm(y, self.proj.w, out_dtype=F32) if self.proj is not None else y.float()
        if record:
            ctx.update(cls=cls, sf=sf, S=chips.shape[2])
            self._ctx = ctx
        return e

    __call__ = forward

    # ------------------------------------------------------------------ backward (input gradient only)
    def backward(self, d_e, gscale, out=None):
        """d_e [N,out_dim] fp32 = dL/d(embedding).  Returns dL/d(chips) [N,3,S,S] fp32 (accumulated into ``out`` if given).
        Intermediate fp16 gradients carry the power-of-two ``gscale``; it is removed when the result is written."""
        cfg, c = self.config, self._ctx
        N = c["N"]
        D, H, d, T, Tp = cfg.hidden_size, self.H, self.d, self.T, self.Tp
        g16 = ops.to_f16(d_e.contiguous(), gscale)
        dy = ops.gemm(g16, self.proj.wT) if self.proj is not None else g16
        dcls = ops.layernorm_bwd(c["cls"], dy, self.final_ln.gamma, c["sf"])
        dx = torch.zeros((N, Tp, D), dtype=F16, device=self.device)
        dx[:, 0] = dcls
        dx = dx.view(N * Tp, D)
        for L, s in zip(reversed(self.layers), reversed(c["layers"])):
            dm = ops.gemm(dx, L["fc2"].wT)
            dz = ops.act_bwd(s["z"], dm, "gelu")
            dn2 = ops.gemm(dz, L["fc1"].wT)
            dh1 = ops.layernorm_bwd(s["h1"], dn2, L["ln2"].gamma, s["s2"], add=dx)
            da = ops.gemm(dh1, L["o"].wT)
            dq, dk, dv = ops.attn_bwd(s["q"], s["k"], s["v"], s["a"], da, s["lse"], N, H, Tp, T, d, kv_rows=Tp)
            dn1 = ops.gemm(dq, L["q"].wT)
            dn1 = ops.gemm(dk, L["k"].wT, residual=dn1)
            dn1 = ops.gemm(dv, L["v"].wT, residual=dn1)
            dx = ops.layernorm_bwd(s["x"], dn1, L["ln1"].gamma, s["s1"], add=dh1)
        if self.pre_ln is not None:
            dx = ops.layernorm_bwd(c["x0"], dx, self.pre_ln.gamma, c["s0"])
        dpe = dx.view(N, Tp, D)[:, 1:T]                                                        # rows of the patch tokens
        dpatches = ops.gemm_batched_from(dpe, self.patch.wT, N, self.g * self.g)               # [N*g*g, Kp]
        self._ctx = None
        return ops.patchify_bwd(dpatches, self.std, N, c["S"], cfg.patch_size, 1.0 / gscale, out=out)


def feature_loss_and_grad(e, target, weights):
    """``loss = 1 - <normalize(e), target>`` per image (:1909-1910) and dL/de for ``sum_i weights_i * loss_i``.
    e [N,E] fp32 raw embeddings, target [N,E] fp32 (already normalised, constant), weights [N] fp32."""
    nrm = e.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    eh = e / nrm
    loss = 1.0 - (eh * target).sum(dim=-1)
    g = -weights[:, None] * target
    de = (g - eh * (eh * g).sum(dim=-1, keepdim=True)) / nrm
    return loss, de
