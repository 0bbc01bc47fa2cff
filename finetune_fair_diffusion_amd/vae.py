"""MI355X-native mirror of the decoder half of diffusers ``AutoencoderKL`` for the reference's
``vae.decode(latents.to(vae.dtype)).sample.clamp(-1, 1)`` (exp-1-debias-gender/1-main-debias.py
:1058-1059, :1133-1134).  Forward (optionally recording) and an explicit data-gradient backward
(all VAE weights are frozen).  The single-head 512-wide mid-block attention runs once per image,
so it uses strided-batch MFMA GEMMs + a row-softmax kernel instead of the fused U-Net kernel.
"""
import torch

from . import ops
from .layers import F16, F32, Conv3x3, Linear, Norm, ResnetBlock
from .weights import VAEConfig, vae_param_shapes


class _Out:
    def __init__(self, sample):
        self.sample = sample


class VAEAttention:
    def __init__(self, sd, p, dev, C, groups):
        self.C, self.groups = C, groups
        self.norm = Norm(sd, p + "group_norm", dev)
        self.q, self.k, self.v, self.o = (Linear(sd, p + n, dev) for n in ("to_q", "to_k", "to_v", "to_out.0"))

    def forward(self, x, B, H, W, ctx=None):
        T, C = H * W, self.C
        scale = C ** -0.5
        g, st = ops.groupnorm(x, None, B, T, self.groups, 1e-6, self.norm.gamma, self.norm.beta, False)
        q = ops.gemm(g, self.q.w, bias=self.q.bias)
        k = ops.gemm(g, self.k.w, bias=self.k.bias)
        v = ops.gemm(g, self.v.w, bias=self.v.bias)
        S = ops.bgemm(q.view(B, T, C), k.view(B, T, C), alpha=scale)
        P = ops.softmax_rows(S)
        vt = ops.transpose_btc(v, B, T, C, T)
        O = ops.bgemm(P, vt).view(B * T, C)
        out = ops.gemm(O, self.o.w, bias=self.o.bias, residual=x)
        if ctx is not None:
            ctx.append(dict(x=x, st=st, q=q, k=k, v=v, P=P, scale=scale))
        return out

    def backward(self, d_out, B, H, W, c):
        T, C = H * W, self.C
        scale = c["scale"]
        dO = ops.gemm(d_out, self.o.wT)
        P = c["P"]
        dP = ops.bgemm(dO.view(B, T, C), c["v"].view(B, T, C))
        dS = ops.softmax_rows_bwd(P, dP, 1.0)
        kt = ops.transpose_btc(c["k"], B, T, C, T)
        dq = ops.bgemm(dS, kt, alpha=scale).view(B * T, C)
        dSt = ops.transpose_btc(dS.view(B * T, T), B, T, T, T)
        qt = ops.transpose_btc(c["q"], B, T, C, T)
        dk = ops.bgemm(dSt, qt, alpha=scale).view(B * T, C)
        Pt = ops.transpose_btc(P.view(B * T, T), B, T, T, T)
        dOt = ops.transpose_btc(dO, B, T, C, T)
        dv = ops.bgemm(Pt, dOt).view(B * T, C)
        dg = ops.gemm(dq, self.q.wT)
        dg = ops.gemm(dk, self.k.wT, residual=dg)
        dg = ops.gemm(dv, self.v.wT, residual=dg)
        dx, _ = ops.groupnorm_bwd(c["x"], None, dg, B, T, self.groups, c["st"], self.norm.gamma, self.norm.beta, False, add1=d_out)
        return dx


class AutoencoderKL:
    """Decoder-only mirror (the encoder is never used on the path)."""

    def __init__(self, cfg: VAEConfig, state_dict, device):
        self.config, self.device = cfg, device
        sd, dev = state_dict, device
        missing = [k for k in vae_param_shapes(cfg) if k not in sd]
        if missing:
            raise KeyError(f"VAE state_dict is missing {len(missing)} tensors, e.g. {missing[:3]}")
        boc, g, n, L = cfg.block_out_channels, cfg.norm_num_groups, cfg.layers_per_block, cfg.latent_channels
        self.dtype = F16
        self.pq_w = sd["post_quant_conv.weight"].to(dev, F32).permute(2, 3, 1, 0).reshape(L, L).contiguous()
        self.pq_b = sd["post_quant_conv.bias"].to(dev, F32).contiguous()
        c = boc[-1]
        self.conv_in = Conv3x3(sd, "decoder.conv_in", dev)
        self.conv_in_w = sd["decoder.conv_in.weight"].to(dev, F32).permute(2, 3, 1, 0).reshape(9 * L, c).contiguous()
        self.mid_res = [ResnetBlock(sd, "decoder.mid_block.resnets.0.", dev, g, 1e-6, has_temb=False),
                        ResnetBlock(sd, "decoder.mid_block.resnets.1.", dev, g, 1e-6, has_temb=False)]
        self.mid_attn = VAEAttention(sd, "decoder.mid_block.attentions.0.", dev, c, g)
        self.ups = []
        rev = list(reversed(boc))
        for i in range(len(rev)):
            self.ups.append(dict(res=[ResnetBlock(sd, f"decoder.up_blocks.{i}.resnets.{j}.", dev, g, 1e-6, has_temb=False) for j in range(n + 1)],
                                 up=Conv3x3(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", dev) if i != len(rev) - 1 else None))
        self.norm_out = Norm(sd, "decoder.conv_norm_out", dev)
        self.conv_out = Conv3x3(sd, "decoder.conv_out", dev)
        wo = sd["decoder.conv_out.weight"].to(dev, F32)
        self.conv_out_wd = wo.flip(2, 3).permute(2, 3, 0, 1).reshape(9 * cfg.out_channels, boc[0]).contiguous()
        self._ctx = None

    def decode_images(self, z, record=False):
        """z: [N,4,h,w] fp32 (already divided by scaling_factor).  Returns clamp(decode(z),-1,1) as [N,3,8h,8w] fp16."""
        cfg = self.config
        B, L, H, W = z.shape
        ctx = [] if record else None
        z16 = ops.to_f16(z.contiguous())
        pq, _, _ = ops.conv_small_cin(z16, self.pq_w, self.pq_b, B, H, W, L, L, 1, 1, nchw=True)
        x, _, _ = ops.conv_small_cin(pq, self.conv_in_w, self.conv_in.bias, B, H, W, L, self.conv_in.cout, 3, 1, nchw=False)
        x = self.mid_res[0].forward(x, None, B, H, W, None, ctx)
        x = self.mid_attn.forward(x, B, H, W, ctx)
        x = self.mid_res[1].forward(x, None, B, H, W, None, ctx)
        for blk in self.ups:
            for r in blk["res"]:
                x = r.forward(x, None, B, H, W, None, ctx)
            if blk["up"] is not None:
                x, H, W = ops.conv_up2(x, blk["up"], B, H, W)
        g, st = ops.groupnorm(x, None, B, H * W, cfg.norm_num_groups, 1e-6, self.norm_out.gamma, self.norm_out.beta, True)
        y, _, _ = ops.conv3x3(g, self.conv_out.wk, B, H, W, bias=self.conv_out.bias)
        img = ops.nhwc_to_nchw(y, B, H * W, cfg.out_channels, out_dtype=F16, lo=-1.0, hi=1.0).view(B, cfg.out_channels, H, W)
        if record:
            self._ctx = dict(blocks=ctx, x_out=x, st_out=st, pre=y, B=B, H=H, W=W, h=z.shape[2], w=z.shape[3])
        return img

    def to(self, *a, **k):
        return self

    def requires_grad_(self, flag=False):
        return self

    def decode(self, z):
        """diffusers-style: ``vae.decode(latents).sample`` (un-clamped values are not exposed; the
        reference clamps immediately)."""
        return _Out(self.decode_images(z.to(F32)))

    def backward_images(self, d_img, gscale):
        """d_img: [N,3,H,W] fp32 = dL/d(images).  Returns dL/dz [N,4,h,w] fp32 (un-scaled)."""
        cfg, c = self.config, self._ctx
        B, H, W = c["B"], c["H"], c["W"]
        blocks = c["blocks"]
        dpre = ops.clamp_bwd(c["pre"], (d_img * gscale).contiguous().view(B, cfg.out_channels, H * W), B, H * W, cfg.out_channels)
        dg, _, _ = ops.conv_small_cin(dpre.view(B, cfg.out_channels, H, W), self.conv_out_wd, None, B, H, W, cfg.out_channels,
                                      self.conv_out.cin, 3, 1, nchw=True)
        dx, _ = ops.groupnorm_bwd(c["x_out"], None, dg, B, H * W, cfg.norm_num_groups, c["st_out"], self.norm_out.gamma, self.norm_out.beta, True)
        for blk in reversed(self.ups):
            if blk["up"] is not None:
                H, W = H // 2, W // 2
                dx = ops.conv_up2_bwd(dx, blk["up"], B, H, W)
            for r in reversed(blk["res"]):
                dx, _ = r.backward(dx, B, H, W, blocks.pop())
        dx, _ = self.mid_res[1].backward(dx, B, H, W, blocks.pop())
        dx = self.mid_attn.backward(dx, B, H, W, blocks.pop())
        dx, _ = self.mid_res[0].backward(dx, B, H, W, blocks.pop())
        dpq, _, _ = ops.conv3x3(dx, self.conv_in.wd, B, H, W)  # [M, 4]
        L = cfg.latent_channels
        dz = ops.conv_small_cin_bwd(dpq, self.pq_w, B, H, W, L, L, 1, 1, scale=1.0 / gscale)
        self._ctx = None
        return dz
