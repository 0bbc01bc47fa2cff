"""Weight containers in kernel layout + the forward/backward building blocks shared by the
U-Net, VAE decoder, text encoder and classifier mirrors.  Activations are channels-last fp16
matrices ``[B*H*W, C]``; frozen weights are fp16 (as the reference casts them, :761-763), LoRA
parameters and their gradients are fp32 (:815, :831).  No autograd: every block has an explicit
backward that only produces data gradients (base weights are frozen) and LoRA weight gradients.
"""
import torch

from . import ops

F16, F32 = ops.F16, ops.F32


class Linear:
    """y = x W^T + b.  ``w`` [N,K] fp16, ``bias`` fp32; ``wT`` [K,N] built on first backward use."""

    def __init__(self, sd, name, dev, conv1x1=False):
        w = sd[name + ".weight"]
        if conv1x1:
            w = w.reshape(w.shape[0], w.shape[1])
        self.w = w.to(dev, F16).contiguous()
        b = sd.get(name + ".bias")
        self.bias = b.to(dev, F32).contiguous() if b is not None else None
        self._wT = None

    @property
    def wT(self):
        if self._wT is None:
            self._wT = self.w.t().contiguous()
        return self._wT


class Conv3x3:
    """3x3 conv weights as implicit-GEMM operands: ``wk`` [Cout, 9*Cin] (ky,kx,ci) and, for the data
    gradient, ``wd`` [Cin, 9*Cout] (spatially flipped, in/out swapped)."""

    def __init__(self, sd, name, dev):
        w = sd[name + ".weight"].to(dev, F16)
        self.cout, self.cin = w.shape[0], w.shape[1]
        self.wk = w.permute(0, 2, 3, 1).reshape(self.cout, 9 * self.cin).contiguous()
        self._w = w
        self.bias = sd[name + ".bias"].to(dev, F32).contiguous()
        self._wd = None

    @property
    def wd(self):
        if self._wd is None:
            self._wd = self._w.flip(2, 3).permute(1, 2, 3, 0).reshape(self.cin, 9 * self.cout).contiguous()
        return self._wd

    def _phase_weights(self):
        """conv3x3(nearest_up2(x)) at output pixel (2y+py, 2x+px) only sees the 2x2 low-res pixels (y+dy+py-1, x+dx+px-1): per phase the
        3x3 kernel folds into a 2x2 one, Wp[py,px,dy,dx] = sum_{ky,kx} R[py,dy,ky] R[px,dx,kx] w[ky,kx] (summed in fp32, then fp16)."""
        R = torch.tensor([[[1., 0., 0.], [0., 1., 1.]], [[1., 1., 0.], [0., 0., 1.]]], device=self._w.device)   # [p, d, k]
        return torch.einsum("pdk,qel,nckl->pqdenc", R, R, self._w.float())                                      # [py,px,dy,dx,N,C]

    @property
    def wk_up2p(self):
        """[4*Cout, 4*Cin]: phase-major (py*2+px) forward operand, k = (dy*2+dx)*Cin + c  (FD_CONV_UP2P)."""
        if getattr(self, "_wk_up2p", None) is None:
            wp = self._phase_weights()
            self._wk_up2p = wp.permute(0, 1, 4, 2, 3, 5).reshape(4 * self.cout, 4 * self.cin).to(F16).contiguous()
        return self._wk_up2p

    @property
    def wd_up2p(self):
        """[Cin, 16*Cout]: input-gradient operand, k = (((py*2+px)*2+dy)*2+dx)*Cout + n  (FD_CONV_UP2P_BWD)."""
        if getattr(self, "_wd_up2p", None) is None:
            wp = self._phase_weights()
            self._wd_up2p = wp.permute(5, 0, 1, 2, 3, 4).reshape(self.cin, 16 * self.cout).to(F16).contiguous()
        return self._wd_up2p


class Norm:
    def __init__(self, sd, name, dev):
        self.gamma = sd[name + ".weight"].to(dev, F32).contiguous()
        self.beta = sd[name + ".bias"].to(dev, F32).contiguous()


def rank_pad(r):
    return 8 if r <= 8 else 16 if r <= 16 else 32 if r <= 32 else 64


class LoRAPair:
    """One LoRALinearLayer (down [r,K], up [N,r]) living in a flat fp32 parameter buffer, with 16-bit rank-padded operand copies for the
    MFMA slab -- ``down16`` [rp,K], ``up16`` [N,rp] and their transposes -- allocated ONCE (possibly as strided views into a stacked
    buffer shared with other pairs, ``place``) and rewritten in place after every optimiser step by ``refresh_pairs``."""

    def __init__(self, bank, down_name, up_name):
        self.bank, self.dn, self.un = bank, down_name, up_name
        self.r = bank.shape(down_name)[0]
        self.rp = rank_pad(self.r)
        self.K = bank.shape(down_name)[1]
        self.N = bank.shape(up_name)[0]
        self.down16 = self.downT16 = self.up16 = self.upT16 = None

    def place(self, down16=None, downT16=None, up16=None, upT16=None):
        """Operand storage: the given views (zero-initialised by their owner; unit column stride) or own buffers."""
        dev = self.bank.flat.device
        self.down16 = down16 if down16 is not None else torch.zeros((self.rp, self.K), dtype=F16, device=dev)
        self.downT16 = downT16 if downT16 is not None else torch.zeros((self.K, self.rp), dtype=F16, device=dev)     # [K, rp]
        self.up16 = up16 if up16 is not None else torch.zeros((self.N, self.rp), dtype=F16, device=dev)
        self.upT16 = upT16 if upT16 is not None else torch.zeros((self.rp, self.N), dtype=F16, device=dev)          # [rp, N]
        for t, shp in ((self.down16, (self.rp, self.K)), (self.downT16, (self.K, self.rp)), (self.up16, (self.N, self.rp)), (self.upT16, (self.rp, self.N))):
            assert tuple(t.shape) == shp and t.stride(1) == 1 and t.dtype == F16, (t.shape, shp, t.stride())
        return self

    def refresh(self, scale=1.0):
        refresh_pairs([self], scale)

    def grads(self):
        """Views of this pair's gradients inside the bank's ACCUMULATION buffer (``bank.grad``, or the second buffer while the
        backward of a timestep runs on the side stream, step.py)."""
        return self.bank.view(self.dn, self.bank.accum), self.bank.view(self.un, self.bank.accum)


def refresh_pairs(pairs, scale=1.0):
    """Rewrites the 16-bit operand copies of ``pairs`` from their fp32 parameters: ONE call of ``fd_lora_refresh_multi`` (16 pairs per launch)
    instead of ~8 tiny torch launches per pair (128 pairs in the U-Net: 10 ms of host-bound time per optimiser step before)."""
    import ctypes
    from . import lib as _lib
    pairs = list(pairs)
    if not pairs:
        return
    arr = _lib.LoraRefreshDesc.array(len(pairs))
    for d, p in zip(arr, pairs):
        if p.down16 is None:
            p.place()
        dn, up = p.bank.view(p.dn), p.bank.view(p.un)
        d.down, d.up = dn.data_ptr(), up.data_ptr()
        d.d16, d.ld_d16, d.dT16, d.ld_dT16 = p.down16.data_ptr(), p.down16.stride(0), p.downT16.data_ptr(), p.downT16.stride(0)
        d.u16, d.ld_u16, d.uT16, d.ld_uT16 = p.up16.data_ptr(), p.up16.stride(0), p.upT16.data_ptr(), p.upT16.stride(0)
        d.r, d.rp, d.K, d.N, d.scale = p.r, p.rp, p.K, p.N, scale
    ops._call("fd_lora_refresh_multi", ctypes.byref(arr), len(pairs), ops._stream())


class ParamBank:
    """Flat fp32 buffer holding named LoRA tensors (+ grad/Adam/EMA twins) so that the gradient
    all-reduce, finite check, AdamW and EMA are each ONE launch over one contiguous buffer
    (replaces the per-parameter loops at 1-main-debias.py:1998-2029)."""

    def __init__(self, shapes, dev):
        self.names = list(shapes.keys())
        self._shape = dict(shapes)
        self.offsets = {}
        off = 0
        for n, s in shapes.items():
            num = 1
            for v in s:
                num *= v
            self.offsets[n] = (off, num)
            off += (num + 3) // 4 * 4
        self.numel = off
        self.flat = torch.zeros(off, dtype=F32, device=dev)
        self.grad = torch.zeros(off, dtype=F32, device=dev)
        self.accum = self.grad      # buffer the LoRA weight-gradient kernels add into
        self._grad_alt = {}
        self.exp_avg = torch.zeros(off, dtype=F32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=F32, device=dev)
        self.ema = torch.zeros(off, dtype=F32, device=dev)

    def shape(self, n):
        return self._shape[n]

    def view(self, n, buf=None):
        off, num = self.offsets[n]
        return (self.flat if buf is None else buf)[off:off + num].view(self._shape[n])

    def grad_view(self, n):
        return self.view(n, self.grad)

    def grad_alt(self, k=1):
        """k-th extra accumulation buffer: timesteps whose backward runs on side stream k add here (no two streams update one buffer);
        summed into ``grad`` once per step."""
        if k not in self._grad_alt:
            self._grad_alt[k] = torch.zeros_like(self.grad)
        return self._grad_alt[k]

    def load_state_dict(self, sd, strict=True):
        for n in self.names:
            if n in sd:
                self.view(n).copy_(sd[n].to(self.flat.device, F32))
            elif strict:
                raise KeyError(n)
        self.ema.copy_(self.flat)

    def state_dict(self, ema=False):
        buf = self.ema if ema else self.flat
        return {n: self.view(n, buf).detach().cpu().clone() for n in self.names}


# ------------------------------------------------------------------------------------------ LoRA linear
def lora_linear_fwd(x, lin, lora, t=None, residual=None, act="none", rowbias=None, rows_per_batch=0, ln=None, colscale=None):
    """y = x W^T (+ (x down^T) up^T) + b (+ residual); returns (y, t) with t = x down^T [M, rp].
    ``ln`` = (gamma, beta, eps): y is the triple (y, LayerNorm(y), per-row statistics) of ``ops.gemm(..., ln=...)``.
    ``colscale`` = (factor, columns): see ``ops.gemm`` (pre-scaled attention queries)."""
    if lora is None:
        return ops.gemm(x, lin.w, bias=lin.bias, residual=residual, act=act, ln=ln, colscale=colscale), None
    if t is None:
        t = ops.gemm(x, lora.down16)
    y = ops.gemm(x, lin.w, a2=t, b2=lora.up16, bias=lin.bias, residual=residual, act=act, ln=ln, colscale=colscale)
    return y, t


def lora_linear_bwd(dy, x, t, lin, lora, gscale, residual=None, need_dx=True):
    """dx = dy W (+ (dy up) down) (+ residual); accumulates LoRA grads (unscaled by 1/gscale)."""
    if lora is None:
        return ops.gemm(dy, lin.wT, residual=residual) if need_dx else None
    u = ops.gemm(dy, lora.upT16)  # [M, rp]
    gd, gu = lora.grads()
    ops.lora_wgrad(dy, t, gu, lora.r, 1, lora.r, scale=1.0 / gscale)    # d up[N,r]   = dy^T t
    ops.lora_wgrad(x, u, gd, 1, lora.K, lora.r, scale=1.0 / gscale)      # d down[r,K] = u^T x
    if not need_dx:
        return None
    return ops.gemm(dy, lin.wT, a2=u, b2=lora.downT16, residual=residual)


# ------------------------------------------------------------------------------------------ ResnetBlock2D
class ResnetBlock:
    def __init__(self, sd, p, dev, groups, eps, has_temb=True):
        self.norm1, self.norm2 = Norm(sd, p + "norm1", dev), Norm(sd, p + "norm2", dev)
        self.conv1, self.conv2 = Conv3x3(sd, p + "conv1", dev), Conv3x3(sd, p + "conv2", dev)
        self.shortcut = Linear(sd, p + "conv_shortcut", dev, conv1x1=True) if (p + "conv_shortcut.weight") in sd else None
        self.groups, self.eps = groups, eps
        self.temb_w = sd.get(p + "time_emb_proj.weight") if has_temb else None
        self.temb_b = sd.get(p + "time_emb_proj.bias") if has_temb else None
        self.temb_slice = None  # (start, stop) into the stacked time-projection table

    def forward(self, x, skip, B, H, W, temb_row=None, ctx=None):
        """x [M,C1] (+ skip [M,C2] concatenated on channels). temb_row: [1, Cout] fp16 or None."""
        HW, M = H * W, B * H * W
        g, st1 = ops.groupnorm(x, skip, B, HW, self.groups, self.eps, self.norm1.gamma, self.norm1.beta, True)
        h, _, _ = ops.conv3x3(g, self.conv1.wk, B, H, W, bias=self.conv1.bias, rowbias=temb_row, gn_stats=True)      # feeds norm2
        g2, st2 = ops.groupnorm(h, None, B, HW, self.groups, self.eps, self.norm2.gamma, self.norm2.beta, True)
        if self.shortcut is not None:
            C1 = x.shape[1]
            if skip is None:
                sc = ops.gemm(x, self.shortcut.w, bias=self.shortcut.bias)
            else:
                sc = ops.gemm(x, self.shortcut.w[:, :C1], a2=skip, b2=self.shortcut.w[:, C1:], bias=self.shortcut.bias)
        else:
            sc = x
        out, _, _ = ops.conv3x3(g2, self.conv2.wk, B, H, W, bias=self.conv2.bias, residual=sc, gn_stats=True)      # feeds the next block's norm
        if ctx is not None:
            ctx.append(dict(x=x, skip=skip, st1=st1, h=h, st2=st2))
        return out

    def backward(self, d_out, B, H, W, c, need_dx=True):
        HW = H * W
        dg2, _, _ = ops.conv3x3(d_out, self.conv2.wd, B, H, W)
        dh, _ = ops.groupnorm_bwd(c["h"], None, dg2, B, HW, self.groups, c["st2"], self.norm2.gamma, self.norm2.beta, True)
        if not need_dx:
            return None, None
        dg1, _, _ = ops.conv3x3(dh, self.conv1.wd, B, H, W)
        x, skip = c["x"], c["skip"]
        if self.shortcut is None:
            add1, add2 = d_out, None
        else:
            C1 = x.shape[1]
            add1 = ops.gemm(d_out, self.shortcut.wT[:C1])
            add2 = ops.gemm(d_out, self.shortcut.wT[C1:]) if skip is not None else None
        return ops.groupnorm_bwd(x, skip, dg1, B, HW, self.groups, c["st1"], self.norm1.gamma, self.norm1.beta, True, add1=add1, add2=add2)
