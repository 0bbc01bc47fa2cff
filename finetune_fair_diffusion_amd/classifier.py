"""MI355X-native mirror of torchvision ``mobilenet_v3_large`` (eval mode, BatchNorm folded into the
convolutions) -- the face-attribute classifier of exp-1-debias-gender/1-main-debias.py:929-935 --
with the forward that scores face chips (:1369-1371) and the explicit data-gradient backward that
carries dL/dlogits back to the face chips (the reference gets it from autograd through the frozen
classifier).  Pointwise convolutions run on the MFMA GEMM, depthwise / squeeze-excite pieces on
the direct kernels of smallconv.hip.
"""
import torch

from . import ops
from .layers import F16, F32
from .weights import MBV3_SETTINGS, make_divisible, mobilenet_param_shapes


def _fold(sd, p, dev):
    """conv (p+'0') followed by BatchNorm (p+'1', eps 1e-3) -> (weight', bias')."""
    w = sd[p + "0.weight"].to(dev, F32)
    g, b = sd[p + "1.weight"].to(dev, F32), sd[p + "1.bias"].to(dev, F32)
    m, v = sd[p + "1.running_mean"].to(dev, F32), sd[p + "1.running_var"].to(dev, F32)
    s = g / torch.sqrt(v + 1e-3)
    return w * s.view(-1, 1, 1, 1), b - m * s


class _PW:  # pointwise conv / linear on the GEMM
    def __init__(self, w, b):
        self.w = w.reshape(w.shape[0], -1).to(F16).contiguous()
        self.bias = b.to(F32).contiguous()
        self.wT = self.w.t().contiguous()


class _DW:
    def __init__(self, w, b, k, stride):
        C = w.shape[0]
        self.w = w.reshape(C, k * k).t().contiguous()  # [k*k, C] fp32
        self.bias = b.contiguous()
        self.k, self.stride = k, stride


class MobileNetV3Large:
    def __init__(self, state_dict, device, num_classes=80):
        sd, dev = state_dict, device
        missing = [k for k in mobilenet_param_shapes(num_classes) if k not in sd]
        if missing:
            raise KeyError(f"classifier state_dict is missing {len(missing)} tensors, e.g. {missing[:3]}")
        self.device, self.num_classes = dev, num_classes
        w, b = _fold(sd, "features.0.", dev)
        self.stem_w = w.permute(2, 3, 1, 0).reshape(27, 16).contiguous()
        self.stem_b = b.contiguous()
        self.blocks = []
        cin = 16
        for i, (k, exp, cout, se, act, s) in enumerate(MBV3_SETTINGS):
            p = f"features.{i + 1}.block."
            j = 0
            blk = dict(k=k, exp=exp, cout=cout, act=act, stride=s, res=(s == 1 and cin == cout), expand=None, se=None)
            if exp != cin:
                blk["expand"] = _PW(*_fold(sd, p + f"{j}.", dev)); j += 1
            w, b = _fold(sd, p + f"{j}.", dev)
            blk["dw"] = _DW(w, b, k, s); j += 1
            if se:
                blk["se"] = (_PW(sd[p + f"{j}.fc1.weight"].to(dev, F32), sd[p + f"{j}.fc1.bias"].to(dev)),
                             _PW(sd[p + f"{j}.fc2.weight"].to(dev, F32), sd[p + f"{j}.fc2.bias"].to(dev)))
                j += 1
            blk["project"] = _PW(*_fold(sd, p + f"{j}.", dev))
            self.blocks.append(blk)
            cin = cout
        self.last = _PW(*_fold(sd, "features.16.", dev))
        self.fc1 = _PW(sd["classifier.0.weight"].to(dev, F32), sd["classifier.0.bias"].to(dev))
        # head padded to a multiple of 8 logits (16-byte GEMM rows); the pad columns are zero weights, sliced off on return
        self.ncls_pad = (num_classes + 7) // 8 * 8
        w2 = torch.zeros(self.ncls_pad, 1280, dtype=F32, device=dev)
        b2 = torch.zeros(self.ncls_pad, dtype=F32, device=dev)
        w2[:num_classes] = sd["classifier.3.weight"].to(dev, F32)
        b2[:num_classes] = sd["classifier.3.bias"].to(dev, F32)
        self.fc2 = _PW(w2, b2)
        self._ctx = None

    def forward(self, chips, record=False):
        """chips: [B,3,S,S] fp16 NCHW.  Returns logits [B, num_classes] fp16."""
        B, _, H, W = chips.shape
        ctx = [] if record else None

        # The recording and the non-recording forward run the SAME kernel sequence (pre-activation rounded to the working dtype, then the
        # activation): the no-grad R1 pass that yields the dynamic targets and the R3 pass whose logits enter the loss are two evaluations of
        # one function in the reference (:1794-1795, :1899-1903) and must agree to the bit here too -- with the activation fused into the
        # GEMM epilogue (applied to the fp32 accumulator) only in the non-recording pass, near-tied probabilities could rank differently in the
        # two passes and flip a target (seen at B = 8 on the random-weight test classifier).  ``record`` only decides what is KEPT.
        def pw(x, lin, act, residual=None):
            """1x1 conv + folded BN (+act)(+residual).  Returns (output, pre-activation or None)."""
            if act == "none":
                return ops.gemm(x, lin.w, bias=lin.bias, act=act, residual=residual), None
            z = ops.gemm(x, lin.w, bias=lin.bias)
            return ops.act_fwd(z, act), (z if record else None)

        z0, H, W = ops.conv_small_cin(chips, self.stem_w, self.stem_b, B, H, W, 3, 16, 3, 2, nchw=True)
        x = ops.act_fwd(z0, "hardswish")
        if record:
            ctx.append(dict(z0=z0, H0=chips.shape[2], W0=chips.shape[3]))
        for blk in self.blocks:
            c = dict(H=H, W=W) if record else None
            inp = x
            act = blk["act"]
            if blk["expand"] is not None:
                x, z = pw(x, blk["expand"], act)
                if record:
                    c["z_exp"] = z
            dw = blk["dw"]
            zd, Ho, Wo = ops.dwconv(x, dw.w, dw.bias, B, H, W, blk["exp"], dw.k, dw.stride, "none")
            x = ops.act_fwd(zd, act)
            if record:
                c["z_dw"] = zd
            H, W = Ho, Wo
            if blk["se"] is not None:
                fc1, fc2 = blk["se"]
                avg = ops.avgpool_hw(x, B, H * W, blk["exp"])
                a1, z1 = pw(avg, fc1, "relu")
                s, z2 = pw(a1, fc2, "hardsigmoid")
                y = ops.scale_channels(x, s, B, H * W, blk["exp"])
                if record:
                    c.update(se_x=x, se_s=s, z1=z1, z2=z2)
                x = y
            x, _ = pw(x, blk["project"], "none", residual=inp if blk["res"] else None)
            if record:
                c.update(Ho=H, Wo=W)
                ctx.append(c)
        x, zl = pw(x, self.last, "hardswish")
        HWl = H * W
        avg = ops.avgpool_hw(x, B, HWl, 960)
        h, zf = pw(avg, self.fc1, "hardswish")
        logits, _ = pw(h, self.fc2, "none")
        if record:
            self._ctx = dict(blocks=ctx, zl=zl, zf=zf, HWl=HWl, B=B)
        return logits[:, :self.num_classes]

    __call__ = forward

    def backward(self, d_logits, gscale, trace=None):
        """d_logits: [B, num_classes] fp32 = dL/dlogits.  Returns dL/dchips [B,3,S,S] fp32."""
        c = self._ctx
        B, blocks = c["B"], c["blocks"]
        dl = d_logits
        if self.ncls_pad != self.num_classes:
            dl = torch.zeros((B, self.ncls_pad), dtype=F32, device=d_logits.device)
            dl[:, :self.num_classes] = d_logits
        d = ops.to_f16(dl.contiguous(), gscale)
        d = ops.gemm(d, self.fc2.wT)
        d = ops.act_bwd(c["zf"], d, "hardswish")
        d = ops.gemm(d, self.fc1.wT)                                   # [B, 960] grad of the pooled features
        d = ops.avgpool_hw_bwd(d, B, c["HWl"], 960)
        d = ops.act_bwd(c["zl"], d, "hardswish")
        d = ops.gemm(d, self.last.wT)
        for blk in reversed(self.blocks):
            cb = blocks.pop()
            H, W, Ho, Wo = cb["H"], cb["W"], cb["Ho"], cb["Wo"]
            d_res = d if blk["res"] else None
            d = ops.gemm(d, blk["project"].wT)                          # grad wrt SE output / dw activation
            if blk["se"] is not None:
                fc1, fc2 = blk["se"]
                dx, ds = ops.scale_channels_bwd(cb["se_x"], cb["se_s"], d, B, Ho * Wo, blk["exp"])
                dz2 = ops.act_bwd(cb["z2"], ds, "hardsigmoid")
                da1 = ops.gemm(dz2, fc2.wT)
                dz1 = ops.act_bwd(cb["z1"], da1, "relu")
                davg = ops.gemm(dz1, fc1.wT)
                d = ops.avgpool_hw_bwd(davg, B, Ho * Wo, blk["exp"], add=dx)
            d = ops.act_bwd(cb["z_dw"], d, blk["act"])
            dw = blk["dw"]
            d = ops.dwconv_bwd(d, dw.w, B, H, W, blk["exp"], dw.k, dw.stride)
            if blk["expand"] is not None:
                d = ops.act_bwd(cb["z_exp"], d, blk["act"])
                d = ops.gemm(d, blk["expand"].wT, residual=d_res)
            elif d_res is not None:
                d = ops.add(d, d_res)
            if trace is not None:
                trace.append(d)
        c0 = blocks.pop()
        d = ops.act_bwd(c0["z0"], d, "hardswish")
        dchips = ops.conv_small_cin_bwd(d, self.stem_w, B, c0["H0"], c0["W0"], 3, 16, 3, 2, scale=1.0 / gscale)
        self._ctx = None
        return dchips
