"""Face-feature network of the face-realism loss term on MI355X: opensphere ``sfnet20`` without norm layers
(opensphere/model/backbone/sfnet.py:123-202, 252-261; loaded at exp-1-debias-gender/1-main-debias.py:968-988, used by
``get_face_feats`` :1176-1190 on 112x112 aligned chips, once on the chip and once on its horizontal mirror).

Channels-last fp16 activations; the 3-channel stem is the direct small-Cin kernel, every other 3x3 convolution the implicit-GEMM
MFMA kernel (stride 2 forward / transposed-stride-2 data gradient as in the U-Net's down-samplers).  ``fc`` consumes the
[7,7,512] map in (y,x,c) order, so its weight columns are permuted once at load from the reference's NCHW flatten order.
Weights are frozen: the backward returns only the gradient w.r.t. the chips.
"""
import torch

from . import ops
from .layers import F16, F32, Conv3x3
from .weights import SFNET20_CHANNELS, SFNET20_LAYERS, sfnet20_param_shapes


class SFNet20:
    def __init__(self, state_dict, device, in_size=112):
        sd, dev = state_dict, device
        missing = [k for k in sfnet20_param_shapes(in_size=in_size) if k not in sd]
        if missing:
            raise KeyError(f"SFNet-20 state_dict is missing {len(missing)} tensors, e.g. {missing[:3]}")
        self.device, self.in_size = dev, in_size
        w = sd["layer1.0.conv1.weight"].float()
        self.stem_w = w.permute(2, 3, 1, 0).reshape(27, w.shape[0]).contiguous().to(dev)          # [k*k*Cin, Cout] fp32
        self.stem_b = sd["layer1.0.conv1.bias"].to(dev, F32).contiguous()
        self.stages = []
        for i, n in enumerate(SFNET20_LAYERS):
            down = Conv3x3(sd, f"layer{i + 1}.0.conv1", dev) if i > 0 else None
            blocks = [(Conv3x3(sd, f"layer{i + 1}.{j}.conv1", dev), Conv3x3(sd, f"layer{i + 1}.{j}.conv2", dev)) for j in range(1, n + 1)]
            self.stages.append((down, blocks))
        C, g = SFNET20_CHANNELS[3], in_size // 16
        fw = sd["fc.weight"].float().view(-1, C, g * g).permute(0, 2, 1).reshape(-1, g * g * C)  # (c,y,x) columns -> (y,x,c)
        self.fc_w = fw.to(dev, F16).contiguous()
        self.fc_wT = self.fc_w.t().contiguous()
        self.fc_b = sd["fc.bias"].to(dev, F32).contiguous()
        self._ctx = None

    def forward(self, chips, record=False):
        """chips [N,3,S,S] fp16 NCHW in [-1,1] -> features [N,512] fp32 (un-normalised, one view)."""
        N, _, H, W = chips.shape
        ctx = [] if record else None
        x, H, W = ops.conv_small_cin(chips.contiguous(), self.stem_w, self.stem_b, N, H, W, 3, self.stem_w.shape[1], 3, 2, nchw=True, act="relu")
        if record:
            ctx.append(("stem", x, H * 2, W * 2))
        for down, blocks in self.stages:
            if down is not None:
                xin, Hin, Win = x, H, W
                x, H, W = ops.conv3x3(x, down.wk, N, H, W, mode=ops.CONV_STRIDE2, bias=down.bias, act="relu")
                if record:
                    ctx.append(("down", down, x, Hin, Win))
            for c1, c2 in blocks:
                h, _, _ = ops.conv3x3(x, c1.wk, N, H, W, bias=c1.bias, act="relu")
                s, _, _ = ops.conv3x3(h, c2.wk, N, H, W, bias=c2.bias, residual=x)
                y = ops.act_fwd(s, "relu")
                if record:
                    ctx.append(("block", c1, c2, h, y, H, W))
                x = y
        f = ops.gemm(x.view(N, -1), self.fc_w, bias=self.fc_b, out_dtype=F32)
        if record:
            self._ctx = dict(ops=ctx, N=N, S=chips.shape[2])
        return f

    __call__ = forward

    def backward(self, d_f, gscale):
        """d_f [N,512] fp32 -> dL/d(chips) [N,3,S,S] fp32.  fp16 gradients carry the power-of-two ``gscale``."""
        c = self._ctx
        N = c["N"]
        dx = ops.gemm(ops.to_f16(d_f.contiguous(), gscale), self.fc_wT)              # [N, 7*7*512] in (y,x,c) order
        dx = dx.view(-1, SFNET20_CHANNELS[3])
        out = None
        for rec in reversed(c["ops"]):
            if rec[0] == "block":
                _, c1, c2, h, y, H, W = rec
                ds = ops.act_bwd(y, dx, "relu")                                      # relu mask from the output: y > 0  <=>  s > 0
                dh, _, _ = ops.conv3x3(ds, c2.wd, N, H, W)
                dh = ops.act_bwd(h, dh, "relu")
                dx, _, _ = ops.conv3x3(dh, c1.wd, N, H, W, residual=ds)              # + identity branch
            elif rec[0] == "down":
                _, down, y, Hin, Win = rec
                d = ops.act_bwd(y, dx, "relu")
                dx, _, _ = ops.conv3x3(d, down.wd, N, (Hin + 1) // 2, (Win + 1) // 2, mode=ops.CONV_TRANS2)
            else:
                _, y, H0, W0 = rec
                d = ops.act_bwd(y, dx, "relu")
                out = ops.conv_small_cin_bwd(d, self.stem_w, N, H0, W0, 3, self.stem_w.shape[1], 3, 2, scale=1.0 / gscale)
        self._ctx = None
        return out


def face_features(net, chips, record=False):
    """get_face_feats (:1176-1190): net(x) + net(flip(x)), fp32, un-normalised here (the caller normalises).  With ``record`` the two
    forward contexts are returned for ``face_features_backward``."""
    flipped = torch.flip(chips, [3]).contiguous()
    f1 = net.forward(chips, record=record)
    c1 = net._ctx
    f2 = net.forward(flipped, record=record)
    c2 = net._ctx
    net._ctx = None
    return f1 + f2, (c1, c2)


def face_features_backward(net, ctxs, d_f, gscale):
    net._ctx = ctxs[0]
    d1 = net.backward(d_f, gscale)
    net._ctx = ctxs[1]
    d2 = net.backward(d_f, gscale)
    return d1 + torch.flip(d2, [3])
