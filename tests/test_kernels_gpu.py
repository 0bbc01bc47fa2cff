"""Kernel-level parity (GPU): every HIP kernel vs a plain fp32 PyTorch statement of the same op,
called through the C-ABI (ops.py -> libfairdiff_hip.so).  Tolerances are fp16-output tolerances:
|err| <= tol * max|ref| with tol stated per test."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def check(name, a, b, tol):
    e = relerr(a, b)
    print(f"[{name}] rel max err {e:.3e} (tol {tol:.1e})")
    assert math.isfinite(e) and e <= tol, f"{name}: {e} > {tol}"


def rnd(*shape, dev, scale=1.0, dtype=torch.float16, seed=None):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed if seed is not None else (hash(shape) & 0xFFFF))
    return (torch.randn(*shape, generator=g) * scale).to(dev).to(dtype)


@pytest.fixture(scope="module")
def ops():
    from finetune_fair_diffusion_amd import ops
    return ops


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 320, 320), (4096, 320, 1280), (1000, 4, 2880), (65, 1280, 40), (2048, 2560, 320)])
def test_gemm_plain(ops, dev, M, N, K):
    a, b = rnd(M, K, dev=dev, seed=1), rnd(N, K, dev=dev, seed=2)
    c = ops.gemm(a, b)
    check(f"gemm {M}x{N}x{K}", c, a.float() @ b.float().t(), 2e-3)


@pytest.mark.parametrize("M,N,K,tile", [(51300, 320, 1280, 256320), (20600, 320, 1288, 128320), (10000, 640, 1032, 128320), (20500, 640, 1032, 256320),
                                          (7000, 480, 1024, 128160), (1300, 1280, 1024, 2128160),
                                          (51300, 128, 1096, 256128), (65536, 320, 320, 256320), (65536, 320, 256, 128064), (1024, 1280, 11520, 8128320), (1000, 320, 5120, 8128160),
                                          (4096, 1280, 11520, 128320), (2048, 1280, 11520, 4128320),
                                          # banded tile order: B slab per 320-wide n-tile 1.3 MB -> bands of 2 n-tiles; 7 n-tiles -> bands 2,2,2,1
                                          (3000, 2240, 2048, 128320), (8200, 2240, 2048, 256320),
                                          (51300, 512, 1096, 256256), (26000, 256, 520, 256128), (205000, 128, 1160, 512128)])
def test_gemm_big_tiles(ops, dev, M, N, K, tile):
    """The 8-wave BK=64 tile variants (incl. M tails and K tails inside a 64-wide k-tile), with the LoRA slab + epilogue."""
    a, b = rnd(M, K, dev=dev, seed=1), rnd(N, K, dev=dev, scale=0.1, seed=2)
    a2, b2 = rnd(M, 8, dev=dev, seed=3), rnd(N, 8, dev=dev, seed=4)
    bias = rnd(N, dev=dev, dtype=torch.float32, seed=5)
    res = rnd(M, N, dev=dev, seed=6)
    import ctypes
    from finetune_fair_diffusion_amd import lib
    d = lib.GemmDesc(); d.M, d.N, d.K, d.batch = M, N, K, 1
    ws = ops.gemm_workspace()
    d.workspace, d.workspace_bytes, d.ldc = ws.data_ptr(), ws.numel() * 4, N
    assert lib.get().fd_gemm_tile(ctypes.byref(d)) == tile
    c = ops.gemm(a, b, a2=a2, b2=b2, bias=bias, residual=res)
    check(f"gemm big {M}x{N}x{K}", c, a.float() @ b.float().t() + a2.float() @ b2.float().t() + bias + res.float(), 2e-3)


@pytest.mark.parametrize("B,H,Cin,Cout", [(16, 32, 640, 640), (5, 16, 1280, 1280), (3, 40, 128, 512), (2, 64, 256, 256), (16, 8, 1280, 1280)])
def test_conv_up2_phase_decomposition(ops, dev, B, H, Cin, Cout):
    """Upsample2D = conv3x3(nearest-up2(x)) as four 2x2-tap phase problems (FD_CONV_UP2P) and its input gradient (FD_CONV_UP2P_BWD),
    against torch fp32 and against the 3x3 gather at the high resolution (FD_CONV_UP2) that it replaces."""
    import torch.nn.functional as F
    from finetune_fair_diffusion_amd.layers import Conv3x3
    x = rnd(B, Cin, H, H, dev=dev, seed=1)
    w = rnd(Cout, Cin, 3, 3, dev=dev, scale=0.03, seed=2)
    bias = rnd(Cout, dev=dev, dtype=torch.float32, seed=3)
    conv = Conv3x3({"c.weight": w, "c.bias": bias}, "c", dev)
    xl = x.permute(0, 2, 3, 1).reshape(B * H * H, Cin).contiguous()
    xr = x.float().requires_grad_(True)
    ref = F.conv2d(F.interpolate(xr, scale_factor=2, mode="nearest"), w.float(), bias, padding=1)
    y, Ho, Wo = ops.conv_up2(xl, conv, B, H, H)
    assert (Ho, Wo) == (2 * H, 2 * H)
    got = y.view(B, Ho, Wo, Cout).permute(0, 3, 1, 2)
    check(f"conv_up2 phases {Cin}->{Cout}@{H}", got, ref, 4e-3)
    old, _, _ = ops.conv3x3(xl, conv.wk, B, H, H, mode=ops.CONV_UP2, bias=conv.bias)
    check("conv_up2 phases vs 3x3 gather", y, old.float(), 4e-3)
    # the phases are written straight into the channels-last result (FD_CONV_UP2PI, the epilogue maps rows); where the kernel has the statistics epilogue
    # (80-column wave tiles) the result carries GroupNorm chunk sums in phase-major order, which the norm consumes (``per``) -- against the two-launch
    # norm on the same tensor
    st = getattr(y, "gn_stats", None)
    if Cout % 320 == 0:
        assert st is not None and len(st) == 3 and st[1] == 32 and st[2] == H * H // 32 and st[0].shape == (4 * B * H * H // 32, Cout // 10, 2)
        gamma = rnd(Cout, dev=dev, dtype=torch.float32, seed=5) * 0.2 + 1
        beta = rnd(Cout, dev=dev, dtype=torch.float32, seed=6) * 0.2
        yn, stn = ops.groupnorm(y, None, B, Ho * Wo, 32, 1e-5, gamma, beta, True)
        yn0, stn0 = ops.groupnorm(y.clone(), None, B, Ho * Wo, 32, 1e-5, gamma, beta, True)
        check("GroupNorm statistics from the phase-major chunk sums", stn, stn0, 2e-5)
        assert float((yn.float() - yn0.float()).abs().max()) <= 2e-3 * float(yn0.float().abs().max())
    g = rnd(B, Cout, Ho, Wo, dev=dev, seed=4)
    ref.backward(g.float())
    gl = g.permute(0, 2, 3, 1).reshape(B * Ho * Wo, Cout).contiguous()
    dx = ops.conv_up2_bwd(gl, conv, B, H, H)
    check(f"conv_up2 phases dgrad {Cin}->{Cout}@{H}", dx.view(B, H, H, Cin).permute(0, 3, 1, 2), xr.grad, 4e-3)


@pytest.mark.parametrize("M,N,K", [(65536, 8, 320), (24576, 24, 320), (16384, 8, 640), (4100, 24, 1280), (1030, 56, 1280), (20000, 8, 768), (4096, 64, 320), (3000, 40, 96)])
def test_gemm_skinny(ops, dev, M, N, K):
    """LoRA down-projection shapes (N = padded rank or three stacked ranks): one wave per 16 rows, K split over the waves of a block.
    A is a column slice of a wider matrix (row stride > K), as the stacked q/k/v down-projection outputs are consumed."""
    import ctypes
    from finetune_fair_diffusion_amd import lib
    wide = rnd(M, K + 64, dev=dev, seed=1)
    a = wide[:, 32:32 + K]
    b = rnd(N, K, dev=dev, scale=0.1, seed=2)
    d = lib.GemmDesc(); d.M, d.N, d.K, d.batch, d.ldc, d.alpha = M, N, K, 1, N, 1.0
    assert lib.get().fd_gemm_tile(ctypes.byref(d)) == 16000 + (N + 15) // 16 * 16
    out = torch.full((M, N + 8), 7.0, dtype=torch.float16, device=dev)
    c = ops.gemm(a, b, out=out[:, :N])
    check(f"gemm skinny {M}x{N}x{K}", c, a.float() @ b.float().t(), 2e-3)
    assert bool((out[:, N:] == 7.0).all())          # nothing written beyond N


@pytest.mark.parametrize("B,H,Cin,Cout", [(16, 64, 320, 320), (14, 64, 64, 128), (13, 32, 128, 640), (3, 136, 128, 256), (2, 168, 64, 512), (1, 456, 64, 128)])
def test_conv3x3_big_tiles(ops, dev, B, H, Cin, Cout):
    x = rnd(B, Cin, H, H, dev=dev, seed=1)
    w = rnd(Cout, Cin, 3, 3, dev=dev, scale=0.05, seed=2)
    bias = rnd(Cout, dev=dev, dtype=torch.float32, seed=3)
    wk = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    y, Ho, Wo = ops.conv3x3(_nhwc(x), wk, B, H, H, bias=bias)
    check("conv big normal", _nchw(y, B, Ho, Wo), F.conv2d(x.float(), w.float(), bias, padding=1), 2e-3)
    y, Ho, Wo = ops.conv3x3(_nhwc(x), wk, B, H, H, mode=ops.CONV_UP2, bias=bias)
    check("conv big up2", _nchw(y, B, Ho, Wo), F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), bias, padding=1), 2e-3)
    y, Ho, Wo = ops.conv3x3(_nhwc(x), wk, B, H, H, mode=ops.CONV_STRIDE2, bias=bias)
    check("conv big stride2", _nchw(y, B, Ho, Wo), F.conv2d(x.float(), w.float(), bias, stride=2, padding=1), 2e-3)
    wd = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, 9 * Cout).contiguous()
    g = rnd(B, Cout, H, H, dev=dev, seed=5)
    dx, Ho, Wo = ops.conv3x3(_nhwc(g), wd, B, H, H, mode=ops.CONV_TRANS2)
    ref = F.conv_transpose2d(g.float(), w.float(), stride=2, padding=1, output_padding=1)
    check("conv big stride2 dgrad", _nchw(dx, B, Ho, Wo), ref, 2e-3)


@pytest.mark.parametrize("B,H,Cin,Cout,kernel", [(16, 64, 64, 320, "conv_halo_kernel<256, 64"), (16, 32, 192, 640, "conv_halo_kernel<256, 32"), (32, 16, 64, 1280, "conv_halo_kernel<256, 16"),
                                                 (2, 64, 64, 640, "conv_halo_kernel<128, 64"), (8, 32, 64, 640, "conv_halo_kernel<128, 32"), (16, 16, 192, 1280, "conv_halo_kernel<128, 16"),
                                                 (3, 64, 320, 640, "conv_halo_kernel<128, 64")])
def test_conv3x3_halo_staged(ops, dev, B, H, Cin, Cout, kernel):
    """Round 6: stride-1 3x3 convolutions of the 64^2 / 32^2 / 16^2 levels with the A operand staged once per 32-channel chunk (image rows + one-pixel halo,
    nine taps = shifted fragment windows, operands by buffer_load ... lds with the hardware's range check supplying the zero padding).  Every tile geometry,
    image borders inside and at the edge of tiles, bias / residual / row-bias epilogues and the GroupNorm-statistics instantiation --
    against fp32 torch, and the dispatch is asserted so that the test cannot pass on the per-tap kernel."""
    import ctypes
    from finetune_fair_diffusion_amd import lib
    x = rnd(B, Cin, H, H, dev=dev, seed=1)
    w = rnd(Cout, Cin, 3, 3, dev=dev, scale=0.05, seed=2)
    bias = rnd(Cout, dev=dev, dtype=torch.float32, seed=3)
    res = rnd(B * H * H, Cout, dev=dev, seed=4)
    rb = rnd(B, Cout, dev=dev, seed=5)
    wk = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    d = lib.GemmDesc(); d.M, d.N, d.K, d.batch, d.conv, d.conv_mode, d.Bn, d.H, d.W, d.Cin, d.Ho, d.Wo = B * H * H, Cout, 9 * Cin, 1, 1, 0, B, H, H, Cin, H, H
    ws = ops.gemm_workspace(); d.workspace, d.workspace_bytes, d.ldc, d.lda, d.ldb = ws.data_ptr(), ws.numel() * 4, Cout, Cin, 9 * Cin
    buf = ctypes.create_string_buffer(128)
    lib.get().fd_gemm_kernel_name(ctypes.byref(d), buf, 128)
    assert buf.value.decode().startswith(kernel), buf.value
    ref = F.conv2d(x.float(), w.float(), bias, padding=1)
    y, _, _ = ops.conv3x3(_nhwc(x), wk, B, H, H, bias=bias)
    check(f"conv halo {kernel}>", _nchw(y, B, H, H), ref, 2e-3)
    y2, _, _ = ops.conv3x3(_nhwc(x), wk, B, H, H, bias=bias, residual=res, rowbias=rb, gn_stats=True)
    ref2 = ref + rb.float()[:, :, None, None] + _nchw(res.float(), B, H, H)
    check(f"conv halo {kernel}> epilogue", _nchw(y2, B, H, H), ref2, 2e-3)
    st = y2.gn_stats[0]
    sref = _unit_sums(y2)
    assert float(((st.double() - sref).abs() / (sref.abs() + 1.0)).max()) < 2e-5
    # an image whose only non-zero pixels sit on the border: every halo row / column of every tile must come back as exact zeros
    xb = torch.zeros_like(x)
    xb[:, :, 0, :], xb[:, :, -1, :], xb[:, :, :, 0], xb[:, :, :, -1] = x[:, :, 0, :], x[:, :, -1, :], x[:, :, :, 0], x[:, :, :, -1]
    yb, _, _ = ops.conv3x3(_nhwc(xb), wk, B, H, H)
    check(f"conv halo {kernel}> border", _nchw(yb, B, H, H), F.conv2d(xb.float(), w.float(), None, padding=1), 2e-3)


@pytest.mark.parametrize("M,N,K,K2", [(4136, 2560, 320, 8), (2048, 5120, 352, 24), (3300, 2560, 328, 0)])
def test_gemm_pingpong_dense_buffer_operands(ops, dev, M, N, K, K2):
    """Round 6: the dense ping-pong kernel (FF1 at the 64^2 level) fetches its operands by buffer_load ... lds -- rows beyond M must read zeros through the range
    check (an out-of-range lane offset), and so must the K tail of a k-step (K % 32 != 0) and of the LoRA slab (K2 = 8 / 24 of a 32-wide step), whatever the scalar k
    offset added to them.  M tails, K tails, slab tails, against fp32 torch; the dispatch is asserted."""
    import ctypes
    from finetune_fair_diffusion_amd import lib
    a, b = rnd(M, K, dev=dev, seed=1), rnd(N, K, dev=dev, scale=0.1, seed=2)
    a2, b2 = (rnd(M, K2, dev=dev, seed=3), rnd(N, K2, dev=dev, seed=4)) if K2 else (None, None)
    bias, res = rnd(N, dev=dev, dtype=torch.float32, seed=5), rnd(M, N, dev=dev, seed=6)
    d = lib.GemmDesc(); d.M, d.N, d.K, d.K2, d.batch, d.ldc, d.lda, d.ldb = M, N, K, K2, 1, N, K, K
    ws = ops.gemm_workspace(); d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    if K2:
        d.A2, d.B2, d.lda2, d.ldb2 = a2.data_ptr(), b2.data_ptr(), K2, K2
    buf = ctypes.create_string_buffer(128)
    lib.get().fd_gemm_kernel_name(ctypes.byref(d), buf, 128)
    assert buf.value.decode().startswith("gemm_pp_kernel<256, 0"), buf.value
    # poisoned neighbours: the operands sit inside larger buffers whose other bytes are huge, so a lane that read past its row / slab would show
    wide = torch.full((M + 8, K + 64), 6e4, dtype=torch.float16, device=dev)
    wide[:M, :K] = a
    c = ops.gemm(wide[:M, :K], b, a2=a2, b2=b2, bias=bias, residual=res)
    ref = a.float() @ b.float().t() + bias + res.float()
    if K2:
        ref = ref + a2.float() @ b2.float().t()
    check(f"gemm pp dense {M}x{N}x{K}+{K2}", c, ref, 2e-3)


def test_gemm_epilogue_and_lora_slab(ops, dev):
    M, N, K, R = 777, 640, 320, 8
    a, b = rnd(M, K, dev=dev, seed=1), rnd(N, K, dev=dev, scale=0.1, seed=2)
    a2, b2 = rnd(M, R, dev=dev, seed=3), rnd(N, R, dev=dev, seed=4)
    bias = rnd(N, dev=dev, dtype=torch.float32, seed=5)
    res = rnd(M, N, dev=dev, seed=6)
    rb = rnd(7, N, dev=dev, seed=7)
    ref = a.float() @ b.float().t() + a2.float() @ b2.float().t() + bias
    ref = ref + rb.float().repeat_interleave(111, 0)
    ref = F.silu(ref) + res.float()
    c = ops.gemm(a, b, a2=a2, b2=b2, bias=bias, rowbias=rb, rows_per_batch=111, residual=res, act="silu")
    check("gemm epilogue", c, ref, 2e-3)
    c32 = ops.gemm(a, b, bias=bias, alpha=0.5, out_dtype=torch.float32, act="gelu")
    check("gemm f32 out gelu", c32, F.gelu(0.5 * (a.float() @ b.float().t()) + bias), 1e-3)
    c3 = ops.gemm(a, b, act="quick_gelu")
    z = a.float() @ b.float().t()
    check("gemm quick_gelu", c3, z * torch.sigmoid(1.702 * z), 2e-3)


def test_bgemm(ops, dev):
    a, b = rnd(6, 200, 64, dev=dev, seed=1), rnd(6, 136, 64, dev=dev, seed=2)
    check("bgemm", ops.bgemm(a, b, alpha=0.25), 0.25 * torch.einsum("zmk,znk->zmn", a.float(), b.float()), 2e-3)


def _nhwc(x):  # [B,C,H,W] -> [B*H*W, C]
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def _nchw(y, B, H, W):
    return y.reshape(B, H, W, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("B,H,Cin,Cout", [(2, 16, 64, 96), (3, 8, 320, 320), (1, 32, 32, 4)])
def test_conv3x3_modes(ops, dev, B, H, Cin, Cout):
    x = rnd(B, Cin, H, H, dev=dev, seed=1)
    w = rnd(Cout, Cin, 3, 3, dev=dev, scale=0.05, seed=2)
    bias = rnd(Cout, dev=dev, dtype=torch.float32, seed=3)
    wk = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    y, Ho, Wo = ops.conv3x3(_nhwc(x), wk, B, H, H, bias=bias)
    check("conv normal", _nchw(y, B, Ho, Wo), F.conv2d(x.float(), w.float(), bias, padding=1), 2e-3)
    y, Ho, Wo = ops.conv3x3(_nhwc(x), wk, B, H, H, mode=ops.CONV_STRIDE2, bias=bias)
    check("conv stride2", _nchw(y, B, Ho, Wo), F.conv2d(x.float(), w.float(), bias, stride=2, padding=1), 2e-3)
    y, Ho, Wo = ops.conv3x3(_nhwc(x), wk, B, H, H, mode=ops.CONV_UP2, bias=bias)
    check("conv up2", _nchw(y, B, Ho, Wo), F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), bias, padding=1), 2e-3)
    # data gradients: flipped/transposed weights [Cin, (ky,kx,co)]
    wd = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, 9 * Cout).contiguous()
    if Cout % 32 == 0:
        xx = x.float().requires_grad_(True)
        o = F.conv2d(xx, w.float(), None, padding=1)
        g = rnd(*o.shape, dev=dev, seed=4)
        o.backward(g.float())
        dx, _, _ = ops.conv3x3(_nhwc(g), wd, B, H, H)
        check("conv dgrad", _nchw(dx, B, H, H), xx.grad, 2e-3)
        xx = x.float().requires_grad_(True)
        o = F.conv2d(xx, w.float(), None, stride=2, padding=1)
        g = rnd(*o.shape, dev=dev, seed=5)
        o.backward(g.float())
        dx, Ho, Wo = ops.conv3x3(_nhwc(g), wd, B, H // 2, H // 2, mode=ops.CONV_TRANS2)
        check("conv stride2 dgrad", _nchw(dx, B, Ho, Wo), xx.grad, 2e-3)


@pytest.mark.parametrize("B,HW,C1,C2,silu", [(2, 256, 320, 0, True), (3, 64, 1280, 640, True), (2, 1024, 128, 0, False), (2, 100, 640, 320, False),
                                             # the U-Net's own small-map shapes (single-launch kernels: 32^2 keeps 21 row vectors per thread in registers)
                                             (2, 1024, 640, 0, True), (2, 1024, 1280, 0, True), (1, 1024, 320, 0, False), (2, 256, 1280, 1280, True),
                                             (2, 64, 1280, 1280, True), (2, 256, 1280, 640, True),
                                             # ... and shapes that stay on the two-launch path (64^2; 1920 channels at 32^2: 61 vectors per thread)
                                             (1, 4096, 320, 0, True), (1, 1024, 1280, 640, True)])
def test_groupnorm(ops, dev, B, HW, C1, C2, silu):
    G, eps = 32, 1e-5
    x1 = rnd(B * HW, C1, dev=dev, seed=1) * 2 + 0.5
    x2 = (rnd(B * HW, C2, dev=dev, seed=2) - 0.3) if C2 else None
    C = C1 + C2
    gamma = rnd(C, dev=dev, dtype=torch.float32, seed=3) * 0.2 + 1
    beta = rnd(C, dev=dev, dtype=torch.float32, seed=4) * 0.2
    xc = (torch.cat([x1, x2], 1) if C2 else x1).float().reshape(B, HW, C).permute(0, 2, 1).requires_grad_(True)
    ref = F.group_norm(xc, G, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    y, st = ops.groupnorm(x1, x2, B, HW, G, eps, gamma, beta, silu)
    check("groupnorm fwd", y.reshape(B, HW, C).permute(0, 2, 1), ref, 2e-3)
    dy = rnd(B * HW, C, dev=dev, seed=5)
    ref.backward(dy.float().reshape(B, HW, C).permute(0, 2, 1))
    add1 = rnd(B * HW, C1, dev=dev, seed=6)
    dx1, dx2 = ops.groupnorm_bwd(x1, x2, dy, B, HW, G, st, gamma, beta, silu, add1=add1)
    gref = xc.grad.permute(0, 2, 1).reshape(B * HW, C)
    check("groupnorm bwd dx1", dx1, gref[:, :C1] + add1.float(), 3e-3)
    if C2:
        check("groupnorm bwd dx2", dx2, gref[:, C1:], 3e-3)


@pytest.mark.parametrize("HW,C1,C2", [(256, 1280, 0), (1024, 640, 0), (64, 1280, 1280), (4096, 320, 0)])
def test_groupnorm_is_batch_invariant_and_reproducible(ops, dev, HW, C1, C2):
    """A sample's GroupNorm (forward, statistics, backward) must not depend on what it is batched with -- the CFG prefix path evaluates N
    samples where the duplicated batch has 2N, and R3 consumes R1's recorded forward -- nor vary from run to run (fixed reduction order)."""
    G, B = 32, 3
    C = C1 + C2
    x1 = rnd(B * HW, C1, dev=dev, seed=1) * 2 + 0.5
    x2 = (rnd(B * HW, C2, dev=dev, seed=2) - 0.3) if C2 else None
    gamma = rnd(C, dev=dev, dtype=torch.float32, seed=3) * 0.2 + 1
    beta = rnd(C, dev=dev, dtype=torch.float32, seed=4) * 0.2
    dy = rnd(B * HW, C, dev=dev, seed=5)
    y, st = ops.groupnorm(x1, x2, B, HW, G, 1e-5, gamma, beta, True)
    dx1, dx2 = ops.groupnorm_bwd(x1, x2, dy, B, HW, G, st, gamma, beta, True)
    y2, st2 = ops.groupnorm(x1, x2, B, HW, G, 1e-5, gamma, beta, True)
    assert torch.equal(y, y2) and torch.equal(st, st2)
    assert torch.equal(dx1, ops.groupnorm_bwd(x1, x2, dy, B, HW, G, st, gamma, beta, True)[0])
    for b in range(B):
        sl = slice(b * HW, (b + 1) * HW)
        yb, stb = ops.groupnorm(x1[sl].contiguous(), x2[sl].contiguous() if C2 else None, 1, HW, G, 1e-5, gamma, beta, True)
        assert torch.equal(yb, y[sl]) and torch.equal(stb[0], st[b])
        d1, d2 = ops.groupnorm_bwd(x1[sl].contiguous(), x2[sl].contiguous() if C2 else None, dy[sl].contiguous(), 1, HW, G, stb, gamma, beta, True)
        assert torch.equal(d1, dx1[sl]) and (not C2 or torch.equal(d2, dx2[sl]))


def _unit_sums(c, rows=32):
    """fp64 statement of fd_gemm_desc.gn_stats: per 32-row chunk and 10-channel unit the (sum, sum of squares) of the stored values."""
    M, N = c.shape
    pad = (-M) % rows
    x = torch.cat([c.double(), torch.zeros(pad, N, dtype=torch.float64, device=c.device)]).reshape((M + pad) // rows, rows, N // 10, 10)
    return torch.stack([x.sum((1, 3)), (x * x).sum((1, 3))], -1)


@pytest.mark.parametrize("kind,shape", [("dense", (32768, 320, 320)), ("dense", (51300, 320, 1280)), ("dense", (12800, 640, 640)), ("dense", (7000, 480, 1024)),
                                        ("conv", (16, 64, 320, 320)), ("conv", (3, 64, 320, 640)), ("conv", (16, 16, 1280, 1280)), ("conv2", (16, 64, 320, 320))])
def test_gemm_groupnorm_statistics_epilogue(ops, dev, kind, shape):
    """fd_gemm_desc.gn_stats: the producer of a GroupNorm's input leaves per-chunk sums of the values it STORED; the output
    itself is bit-identical to the launch without statistics.  Covers the lockstep 256x320 / 128x320 / 128x160 tiles, the ping-pong convolutions,
    the stride-2 gather, residual / bias / row-bias epilogues and an M tail that ends inside a chunk."""
    if kind == "dense":
        M, N, K = shape
        a, b = rnd(M, K, dev=dev, seed=1), rnd(N, K, dev=dev, scale=0.1, seed=2)
        bias, res = rnd(N, dev=dev, dtype=torch.float32, seed=5), rnd(M, N, dev=dev, seed=6)
        run = lambda st: ops.gemm(a, b, bias=bias, residual=res, gn_stats=st)
    else:
        B, H, Cin, Cout = shape
        x, w = rnd(B * H * H, Cin, dev=dev, seed=1), rnd(Cout, 9 * Cin, dev=dev, scale=0.05, seed=2)
        bias, rb = rnd(Cout, dev=dev, dtype=torch.float32, seed=3), rnd(1, Cout, dev=dev, seed=4)
        mode = ops.CONV_STRIDE2 if kind == "conv2" else ops.CONV_NORMAL
        run = lambda st: ops.conv3x3(x, w, B, H, H, mode=mode, bias=bias, rowbias=rb, gn_stats=st)[0]
    plain, c = run(False), run(True)
    assert torch.equal(plain, c)
    assert getattr(plain, "gn_stats", None) is None and c.gn_stats[1] == 32
    st = c.gn_stats[0]
    ref = _unit_sums(c)
    assert st.shape == ref.shape
    err = float(((st.double() - ref).abs() / (ref.abs() + 1.0)).max())
    print(f"[gn_stats {kind} {shape}] max err {err:.2e}")
    assert err < 2e-5
    c2 = run(True)
    assert torch.equal(c2.gn_stats[0], st)          # fixed order


def test_gemm_groupnorm_statistics_do_not_depend_on_the_tile_policy(ops, dev):
    """The chunks are formed by one canonical procedure: the first rows of a big launch (256x320 tiles) and a small launch over those rows alone
    (128x320 tiles) carry bit-identical statistics -- the CFG-pair prefix evaluates N samples where the duplicated batch has 2N."""
    import ctypes
    from finetune_fair_diffusion_amd import lib
    M, Ms, N, K = 32768, 8192, 640, 640
    a, b = rnd(M, K, dev=dev, seed=1), rnd(N, K, dev=dev, scale=0.1, seed=2)
    res = rnd(M, N, dev=dev, seed=3)
    tiles = []
    for m in (M, Ms):
        d = lib.GemmDesc(); d.M, d.N, d.K, d.batch, d.ldc = m, N, K, 1, N
        ws = ops.gemm_workspace()
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        tiles.append(lib.get().fd_gemm_tile(ctypes.byref(d)))
    assert tiles == [256320, 128320], tiles
    big = ops.gemm(a, b, residual=res, gn_stats=True)
    small = ops.gemm(a[:Ms], b, residual=res[:Ms], gn_stats=True)
    assert torch.equal(big[:Ms], small)
    assert torch.equal(big.gn_stats[0][:Ms // 32], small.gn_stats[0])


@pytest.mark.parametrize("B,HW,C1,C2,silu", [(4, 4096, 320, 0, True), (16, 1024, 640, 320, True), (4, 4096, 640, 320, False), (16, 256, 1280, 1280, True),
                                             (2, 16384, 320, 0, True), (2, 16384, 640, 320, True)])
def test_groupnorm_from_producer_statistics(ops, dev, B, HW, C1, C2, silu):
    """fd_groupnorm_fwd_stats: the apply pass alone, statistics assembled from the producers' chunk sums (two producers for a channel
    concatenation whose 30- / 80-channel groups straddle the sources).  Against torch, against the two-launch form, per sample vs batched."""
    G, eps, M = 32, 1e-5, B * HW
    K = 320
    a1, a2 = rnd(M, K, dev=dev, seed=1), rnd(M, K, dev=dev, seed=2)
    b1, b2 = rnd(C1, K, dev=dev, scale=0.1, seed=3), (rnd(C2, K, dev=dev, scale=0.1, seed=4) if C2 else None)
    res1 = rnd(M, C1, dev=dev, seed=5)
    x1 = ops.gemm(a1, b1, residual=res1, gn_stats=True)
    x2 = ops.gemm(a2, b2, gn_stats=True) if C2 else None
    assert x1.gn_stats is not None and (x2 is None or x2.gn_stats is not None)
    C = C1 + C2
    gamma = rnd(C, dev=dev, dtype=torch.float32, seed=6) * 0.2 + 1
    beta = rnd(C, dev=dev, dtype=torch.float32, seed=7) * 0.2
    xc = (torch.cat([x1, x2], 1) if C2 else x1).float().reshape(B, HW, C).permute(0, 2, 1)
    ref = F.group_norm(xc, G, gamma, beta, eps)
    ref = F.silu(ref) if silu else ref
    y, st = ops.groupnorm(x1, x2, B, HW, G, eps, gamma, beta, silu)
    check("groupnorm from producer statistics", y.reshape(B, HW, C).permute(0, 2, 1), ref, 2e-3)
    y0, st0 = ops.groupnorm(x1.clone(), x2.clone() if C2 else None, B, HW, G, eps, gamma, beta, silu)     # clones carry no statistics: reduce + apply
    check("mean / rstd vs the two-launch form", st, st0, 2e-5)
    assert float((y.float() - y0.float()).abs().max()) <= 2e-3 * float(y0.float().abs().max())
    y2, st2 = ops.groupnorm(x1, x2, B, HW, G, eps, gamma, beta, silu)
    assert torch.equal(y, y2) and torch.equal(st, st2)
    for bi in range(B):                                   # a sample alone: other tiles, same statistics, same output
        sl = slice(bi * HW, (bi + 1) * HW)
        x1b = ops.gemm(a1[sl], b1, residual=res1[sl], gn_stats=True)
        x2b = ops.gemm(a2[sl], b2, gn_stats=True) if C2 else None
        if getattr(x1b, "gn_stats", None) is None or (C2 and getattr(x2b, "gn_stats", None) is None):
            continue                                      # the small problem went to a kernel without the statistics epilogue
        yb, stb = ops.groupnorm(x1b, x2b, 1, HW, G, eps, gamma, beta, silu)
        assert torch.equal(yb, y[sl]) and torch.equal(stb[0], st[bi])


@pytest.mark.parametrize("M,C", [(1000, 320), (333, 1280), (64, 768)])
def test_layernorm(ops, dev, M, C):
    x = rnd(M, C, dev=dev, seed=1) * 3 + 1
    gamma = rnd(C, dev=dev, dtype=torch.float32, seed=2) * 0.2 + 1
    beta = rnd(C, dev=dev, dtype=torch.float32, seed=3) * 0.2
    xr = x.float().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gamma, beta, 1e-5)
    y, st = ops.layernorm(x, gamma, beta, 1e-5, save_stats=True)
    check("layernorm fwd", y, ref, 2e-3)
    dy = rnd(M, C, dev=dev, seed=4)
    ref.backward(dy.float())
    add = rnd(M, C, dev=dev, seed=5)
    check("layernorm bwd", ops.layernorm_bwd(x, dy, gamma, st, add=add), xr.grad + add.float(), 3e-3)


def test_elementwise(ops, dev):
    M, Fh = 300, 1280
    proj = rnd(M, 2 * Fh, dev=dev, seed=1)
    pr = proj.float().requires_grad_(True)
    a, g = pr.chunk(2, dim=-1)
    ref = a * F.gelu(g)
    check("geglu fwd", ops.geglu(proj), ref, 2e-3)
    dy = rnd(M, Fh, dev=dev, seed=2)
    ref.backward(dy.float())
    check("geglu bwd", ops.geglu_bwd(proj, dy), pr.grad, 3e-3)
    x = rnd(1003, dev=dev, seed=3) * 3
    for act, fn in [("silu", F.silu), ("relu", F.relu), ("hardswish", F.hardswish), ("hardsigmoid", F.hardsigmoid),
                    ("quick_gelu", lambda t: t * torch.sigmoid(1.702 * t)), ("gelu", F.gelu)]:
        xr = x.float().requires_grad_(True)
        r = fn(xr)
        check(f"act {act}", ops.act_fwd(x, act), r, 2e-3)
        r.backward(torch.ones_like(r) * 0.5)
        check(f"act_bwd {act}", ops.act_bwd(x, torch.full_like(x, 0.5), act), xr.grad, 3e-3)
    a, b = rnd(999, dev=dev, seed=4), rnd(999, dev=dev, seed=5)
    check("add", ops.add(a, b, 0.5, -2.0), 0.5 * a.float() - 2 * b.float(), 2e-3)
    x = rnd(3, 100, 320, dev=dev, seed=6)
    yt = ops.transpose_btc(x.reshape(300, 320), 3, 100, 320)
    assert yt.shape == (3, 320, 104)
    check("transpose", yt[:, :, :100], x.permute(0, 2, 1), 0)
    assert float(yt[:, :, 100:].abs().max()) == 0
    x = rnd(2, 8, 8, 64, dev=dev, seed=7)
    check("downsum", ops.downsum2x2(x.reshape(-1, 64), 2, 4, 4, 64).reshape(2, 4, 4, 64),
          x.float().reshape(2, 4, 2, 4, 2, 64).sum(dim=(2, 4)), 2e-3)
    x = rnd(50, 77, dev=dev, seed=8) * 4
    check("softmax", ops.softmax_rows(x, 0.5), torch.softmax(0.5 * x.float(), -1), 2e-3)
    x = rnd(20, 4096, dev=dev, seed=9) * 4
    p = ops.softmax_rows(x, 0.1)
    check("softmax 4096", p, torch.softmax(0.1 * x.float(), -1), 2e-3)
    dp = rnd(20, 4096, dev=dev, seed=10)
    xr = x.float().requires_grad_(True)
    torch.softmax(0.1 * xr, -1).backward(dp.float())
    check("softmax bwd", ops.softmax_rows_bwd(p, dp, 0.1), xr.grad, 5e-3)
    src = rnd(64, 48, dev=dev, seed=11)
    dst = torch.zeros(64, 80, dtype=torch.float16, device=dev)
    ops.copy_cols(src, dst[:, 32:], 48)
    check("copy_cols", dst[:, 32:], src, 0)


def _attn_ref(q, k, v, H, kv_div=1):
    B, T, C = q.shape
    d = C // H
    kk, vv = k.repeat_interleave(kv_div, 0), v.repeat_interleave(kv_div, 0)
    def sp(t):
        return t.reshape(t.shape[0], t.shape[1], H, d).permute(0, 2, 1, 3)
    s = sp(q) @ sp(kk).transpose(-1, -2) * d ** -0.5
    o = torch.softmax(s, -1) @ sp(vv)
    return o.permute(0, 2, 1, 3).reshape(B, T, C), torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,H,Tq,Tk,d,kv_div", [(2, 8, 1024, 1024, 40, 1), (2, 8, 256, 256, 80, 1), (2, 4, 256, 256, 160, 1),
                                                 (1, 2, 64, 64, 160, 1), (4, 8, 1024, 13, 40, 2), (2, 2, 200, 77, 64, 1),
                                                 (2, 4, 64, 16, 32, 2), (1, 8, 4096, 4096, 40, 1), (2, 4, 100, 50, 16, 1), (1, 2, 130, 130, 128, 1)])
def test_attention_fwd_bwd(ops, dev, B, H, Tq, Tk, d, kv_div):
    """V, K, Q, dO are consumed row-major, as the projections write them, through LDS transpose reads (ds_read_b64_tr_b16).  (Until round 5 the
    library also held the round-1/2 form with transposed copies made by fd_transpose_btc and this test asserted the two bit-identical.)"""
    C = H * d
    Bk = B // kv_div
    q, k, v = rnd(B, Tq, C, dev=dev, seed=1), rnd(Bk, Tk, C, dev=dev, seed=2), rnd(Bk, Tk, C, dev=dev, seed=3)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    oref, lref = _attn_ref(qr, kr, vr, H, kv_div)
    q2, k2, v2 = q.reshape(B * Tq, C), k.reshape(Bk * Tk, C), v.reshape(Bk * Tk, C)
    o, lse = ops.attn_fwd(q2, k2, v2, B, H, Tq, Tk, d, kv_div, need_lse=True)
    check("attn fwd", o.reshape(B, Tq, C), oref, 3e-3)
    check("attn lse", lse, lref, 1e-3)
    do = rnd(B, Tq, C, dev=dev, seed=4)
    oref.backward(do.float())
    dk_acc = torch.zeros(Bk * Tk, C, dtype=torch.float32, device=dev) if kv_div > 1 else None
    dv_acc = torch.zeros_like(dk_acc) if kv_div > 1 else None
    dq, dk, dv = ops.attn_bwd(q2, k2, v2, o, do.reshape(B * Tq, C), lse, B, H, Tq, Tk, d, kv_div, dk_acc=dk_acc, dv_acc=dv_acc)
    check("attn dq", dq.reshape(B, Tq, C), qr.grad, 5e-3)
    check("attn dk", dk.reshape(Bk, Tk, C), kr.grad, 5e-3)
    check("attn dv", dv.reshape(Bk, Tk, C), vr.grad, 5e-3)
    # the deterministic form of shared dK / dV (round 4): per-sample fp32 slabs + fixed-order sum instead of atomics -- same values as the
    # atomics form up to the order of kv_div fp32 additions, BIT-identical between two launches, also at kv_div == 1
    outs = []
    for _ in range(2):
        dko, dvo = torch.full((Bk * Tk, C), 7.0, dtype=torch.float32, device=dev), torch.full((Bk * Tk, C), -3.0, dtype=torch.float32, device=dev)
        dq3, _, _ = ops.attn_bwd(q2, k2, v2, o, do.reshape(B * Tq, C), lse, B, H, Tq, Tk, d, kv_div, dk_out=dko, dv_out=dvo)
        outs.append((dq3, dko, dvo))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])) and torch.equal(outs[0][0], dq)
    check("attn dk (slabs)", outs[0][1].reshape(Bk, Tk, C), kr.grad, 5e-3)
    check("attn dv (slabs)", outs[0][2].reshape(Bk, Tk, C), vr.grad, 5e-3)


@pytest.mark.parametrize("B,H,Tq,Tk,kv_div", [(2, 8, 1024, 1024, 1), (1, 8, 4096, 4096, 1), (4, 8, 1024, 13, 2), (2, 4, 300, 77, 1), (16, 8, 4096, 77, 8)])
def test_attention_with_prescaled_q(ops, dev, B, H, Tq, Tk, kv_div):
    """"Pre-scaled q" (negative ``scale`` of the C-ABI; d = 40): q arrives multiplied by d^-0.5 * log2(e), the kernels take the QK^T accumulator as the
    exponent's argument and carry the softmax reference point / the saved log-sum-exp in the spare contraction slots.  Checked against torch on the
    queries the stored values stand for (q' / factor): forward, LSE, and dq / dk / dv -- dq being the gradient w.r.t. the UNSCALED query, which is what the
    projection's backward consumes; QB = 1 and QB = 2 forwards, shared K / V with the slab form, a partial last key tile, invalid query rows."""
    d = 40
    C, Bk = H * d, B // kv_div
    fac = ops.q_prescale(d)
    assert fac is not None and abs(fac - d ** -0.5 * 1.4426950408889634) < 1e-12
    q, k, v = rnd(B, Tq, C, dev=dev, seed=1), rnd(Bk, Tk, C, dev=dev, seed=2), rnd(Bk, Tk, C, dev=dev, seed=3)
    qp = (q.float() * fac).half()                        # what the projection's epilogue writes (one rounding)
    qr, kr, vr = (qp.float() / fac).requires_grad_(True), k.float().requires_grad_(True), v.float().requires_grad_(True)
    oref, lref = _attn_ref(qr, kr, vr, H, kv_div)
    q2, k2, v2 = qp.reshape(B * Tq, C), k.reshape(Bk * Tk, C), v.reshape(Bk * Tk, C)
    o, lse = ops.attn_fwd(q2, k2, v2, B, H, Tq, Tk, d, kv_div, need_lse=True, prescaled=True)
    check("attn fwd (pre-scaled q)", o.reshape(B, Tq, C), oref, 3e-3)
    check("attn lse (pre-scaled q)", lse, lref, 1e-3)
    o_b, lse_b = ops.attn_fwd(q2, k2, v2, B, H, Tq, Tk, d, kv_div, need_lse=True, prescaled=True)
    assert torch.equal(o, o_b) and torch.equal(lse, lse_b)
    do = rnd(B, Tq, C, dev=dev, seed=4)
    oref.backward(do.float())
    dko, dvo = torch.empty(Bk * Tk, C, dtype=torch.float32, device=dev), torch.empty(Bk * Tk, C, dtype=torch.float32, device=dev)
    dq, _, _ = ops.attn_bwd(q2, k2, v2, o, do.reshape(B * Tq, C), lse, B, H, Tq, Tk, d, kv_div, dk_out=dko, dv_out=dvo, prescaled=True)
    check("attn dq (pre-scaled q)", dq.reshape(B, Tq, C), qr.grad, 5e-3)
    check("attn dk (pre-scaled q)", dko.reshape(Bk, Tk, C), kr.grad, 5e-3)
    check("attn dv (pre-scaled q)", dvo.reshape(Bk, Tk, C), vr.grad, 5e-3)
    if kv_div == 1:
        dq2, dk2, dv2 = ops.attn_bwd(q2, k2, v2, o, do.reshape(B * Tq, C), lse, B, H, Tq, Tk, d, 1, prescaled=True)     # fp16 outputs, no slabs
        assert torch.equal(dq, dq2)
        check("attn dk fp16 (pre-scaled q)", dk2.reshape(Bk, Tk, C), kr.grad, 5e-3)
        check("attn dv fp16 (pre-scaled q)", dv2.reshape(Bk, Tk, C), vr.grad, 5e-3)


@pytest.mark.parametrize("gscale", [1.0, 1e-3, 1e-4])
@pytest.mark.parametrize("prescaled", [False, True])
def test_attention_backward_with_small_upstream_gradients(ops, dev, gscale, prescaled):
    """the d = 40 backward carries -D = -rowsum(dO o O) in three 16-bit pieces of the dO . V^T contraction.  Under loss scaling dO is 1e-3 .. 1e-4 of
    the O(1) values the other kernel tests use and |D| falls below 256 x the smallest normal fp16 number: the round-4 split (all three pieces at scale
    1 / 256) then had a subnormal leading piece -- an absolute error floor on D that is percent-level on dS.  Relative accuracy must not depend on the
    scale of dO (split3_scaled: pieces at the scales 256, 1, 1 / 256, none ever subnormal)."""
    B, H, T, d = 2, 8, 1024, 40
    C = H * d
    fac = ops.q_prescale(d)
    q, k, v = rnd(B, T, C, dev=dev, seed=1), rnd(B, T, C, dev=dev, seed=2), rnd(B, T, C, dev=dev, seed=3)
    qp = (q.float() * fac).half() if prescaled else q
    qr = ((qp.float() / fac) if prescaled else q.float()).requires_grad_(True)
    kr, vr = k.float().requires_grad_(True), v.float().requires_grad_(True)
    oref, _ = _attn_ref(qr, kr, vr, H, 1)
    q2, k2, v2 = qp.reshape(B * T, C), k.reshape(B * T, C), v.reshape(B * T, C)
    o, lse = ops.attn_fwd(q2, k2, v2, B, H, T, T, d, 1, need_lse=True, prescaled=prescaled)
    do = (rnd(B, T, C, dev=dev, seed=4).float() * gscale).half()
    oref.backward(do.float())
    dko, dvo = torch.empty(B * T, C, dtype=torch.float32, device=dev), torch.empty(B * T, C, dtype=torch.float32, device=dev)
    dq, _, _ = ops.attn_bwd(q2, k2, v2, o, do.reshape(B * T, C), lse, B, H, T, T, d, 1, dk_out=dko, dv_out=dvo, prescaled=prescaled)
    # check() is relative to max |reference|: the same 5e-3 as at O(1) upstream gradients
    # dq is an fp16 OUTPUT: at dO x 1e-4 its values (~1e-5) are fp16 subnormals, whose spacing 2^-24 is the floor of any kernel's accuracy there
    check(f"attn dq, dO x {gscale}", dq.reshape(B, T, C), qr.grad, 5e-3 + 4 * 2.0 ** -24 / float(qr.grad.abs().max()))
    check(f"attn dk, dO x {gscale}", dko.reshape(B, T, C), kr.grad, 5e-3)
    check(f"attn dv, dO x {gscale}", dvo.reshape(B, T, C), vr.grad, 5e-3)


def test_attention_prescaled_forward_with_a_very_negative_first_key_tile(ops, dev):
    """in the pre-scaled-q forward the first key tile always moves the softmax reference point; when every score of that tile is below -128 (log2
    domain) the rescale factor exp2(-delta) was +inf and inf * 0 = NaN in the still-empty accumulators.  First 64 keys ~200 below the others."""
    B, H, T, d = 1, 4, 256, 40
    C = H * d
    fac = ops.q_prescale(d)
    q, k, v = rnd(B, T, C, dev=dev, seed=1), rnd(B, T, C, dev=dev, seed=2), rnd(B, T, C, dev=dev, seed=3)
    q, k = q.clone(), k.clone()
    qv, kv = q.view(B, T, H, d), k.view(B, T, H, d)
    qv[..., 0] = 36.0                                     # q_0 * k_0 * d^-0.5 = -36 * 36 * 0.158 = -205 for the first 64 keys, 0 for the others
    kv[..., 0] = 0.0
    kv[:, :64, :, 0] = -36.0
    qp = (q.float() * fac).half()
    oref, lref = _attn_ref((qp.float() / fac), k.float(), v.float(), H, 1)
    for qb_T in (T,):
        o, lse = ops.attn_fwd(qp.reshape(B * T, C), k.reshape(B * T, C), v.reshape(B * T, C), B, H, T, T, d, 1, need_lse=True, prescaled=True)
        assert torch.isfinite(o).all() and torch.isfinite(lse).all()
        check("attn fwd (pre-scaled q, first tile far below the rest)", o.reshape(B, T, C), oref, 3e-3)
        check("attn lse (same)", lse, lref, 1e-3)


@pytest.mark.parametrize("M,N,K,cols", [(4096, 960, 320, 320), (65536, 960, 320, 320), (300, 320, 320, 320), (1024, 3840, 1280, 1280), (2048, 2560, 320, 640)])
def test_gemm_column_scale(ops, dev, M, N, K, cols):
    """fd_gemm_desc.colscale: the first ``cols`` output columns times a factor in the fp32 epilogue, before bias and rounding (the q third of the stacked
    q / k / v projection), on every tile family incl. split-K; the other columns bit-identical to the launch without it."""
    a, b = rnd(M, K, dev=dev, seed=1), rnd(N, K, dev=dev, scale=0.1, seed=2)
    a2, b2 = rnd(M, 8, dev=dev, seed=3), rnd(N, 8, dev=dev, seed=4)
    fac = 0.2280966
    plain = ops.gemm(a, b, a2=a2, b2=b2)
    c = ops.gemm(a, b, a2=a2, b2=b2, colscale=(fac, cols))
    assert torch.equal(c[:, cols:], plain[:, cols:])
    ref = (a.float() @ b.float().t() + a2.float() @ b2.float().t())
    ref[:, :cols] *= fac
    check(f"gemm colscale {M}x{N}x{K}", c, ref, 2e-3)
    bias = rnd(N, dev=dev, dtype=torch.float32, seed=5)
    res = rnd(M, N, dev=dev, seed=6)
    c2 = ops.gemm(a, b, bias=bias, residual=res, colscale=(fac, cols))
    ref2 = a.float() @ b.float().t()
    ref2[:, :cols] *= fac
    check("gemm colscale + bias + residual", c2, ref2 + bias + res.float(), 2e-3)


@pytest.mark.parametrize("C,B,HW,L,kv_div", [(320, 4, 1024, 77, 2), (320, 2, 4096, 77, 1), (640, 4, 256, 77, 2), (640, 2, 1024, 13, 1), (640, 2, 64, 80, 1),
                                              (320, 2, 64, 1, 1)])
def test_cross_attn_block_one_launch(ops, dev, C, B, HW, L, kv_div):
    """fd_cross_attn_block (csrc/crossattn.hip; north_star's named fusion): LayerNorm2 -> attn2.to_q -> attention over the L prompt tokens -> attn2.to_out +
    residual -> LayerNorm3 in ONE launch, against (i) torch fp32 on the same fp16 inputs and (ii) the five separate launches of this library it replaces
    (diffusers BasicTransformerBlock.forward: norm2 / attn2 / norm3; the processor selected at exp-1 main:811-817).  The two paths round q differently (here
    once, pre-scaled; there fp16 first where d != 40), so (ii) is a tolerance, not bit-equality; the LayerNorms and the output epilogue follow the separate
    kernels' arithmetic statement for statement."""
    import torch.nn.functional as F
    H, d, M, Bk = 8, C // 8, B * HW, B // kv_div
    x = rnd(M, C, dev=dev, seed=1)
    g2, b2 = rnd(C, dev=dev, dtype=torch.float32, seed=2) * 0.2 + 1, rnd(C, dev=dev, dtype=torch.float32, seed=3) * 0.2
    g3, b3 = rnd(C, dev=dev, dtype=torch.float32, seed=4) * 0.2 + 1, rnd(C, dev=dev, dtype=torch.float32, seed=5) * 0.2
    wq, wo = rnd(C, C, dev=dev, scale=C ** -0.5, seed=6), rnd(C, C, dev=dev, scale=C ** -0.5, seed=7)
    bo = rnd(C, dev=dev, dtype=torch.float32, seed=8) * 0.1
    k, v = rnd(Bk * L, C, dev=dev, seed=9), rnd(Bk * L, C, dev=dev, seed=10)
    vt = ops.transpose_btc(v, Bk, L, C, ops.CROSS_LP)
    assert ops.cross_block_ok(M, C, H, L, HW)
    y, yn, st, _ = ops.cross_attn_block(x, (g2, b2, 1e-5), wq, k, vt, L, wo, bo, (g3, b3, 1e-5), H, HW, kv_div, need_stats=True)
    # (i) torch fp32
    xf = x.float()
    n2 = F.layer_norm(xf, (C,), g2, b2, 1e-5)
    q = n2 @ wq.float().t()
    o, _ = _attn_ref(q.view(B, HW, C), k.float().view(Bk, L, C), v.float().view(Bk, L, C), H, kv_div)
    yr = o.reshape(M, C) @ wo.float().t() + bo + xf
    ynr = F.layer_norm(yr, (C,), g3, b3, 1e-5)
    check(f"cross block C={C}: y vs fp32", y, yr, 3e-3)
    check(f"cross block C={C}: LayerNorm3(y) vs fp32", yn, ynr, 4e-3)
    check("cross block: LayerNorm3 mean", st[:, 0], yr.mean(-1), 2e-3)
    # (ii) the five launches
    n2s = ops.layernorm(x, g2, b2, 1e-5)
    qs = ops.q_prescale(d)
    q2 = ops.gemm(n2s, wq, colscale=(qs, C) if qs is not None else None)
    o2 = ops.attn_fwd(q2, k, v, B, H, HW, L, d, kv_div, prescaled=qs is not None)
    ys = ops.gemm(o2, wo, bias=bo, residual=x)
    yns, sts = ops.layernorm(ys, g3, b3, 1e-5, save_stats=True)
    check("cross block: y vs the separate launches", y, ys.float(), 2e-3)
    check("cross block: LayerNorm3(y) vs the separate launches", yn, yns.float(), 3e-3)
    frac = float((y != ys).float().mean())
    print(f"cross block C={C} M={M} L={L}: {100 * frac:.2f} % of y differ from the separate launches (by fp16 ulps)")
    # determinism: no atomics, fixed reduction orders
    y2, yn2, _, _ = ops.cross_attn_block(x, (g2, b2, 1e-5), wq, k, vt, L, wo, bo, (g3, b3, 1e-5), H, HW, kv_div)
    assert torch.equal(y, y2) and torch.equal(yn, yn2)


@pytest.mark.parametrize("C,B,HW,L,kv_div,r", [(320, 4, 1024, 77, 2, 4), (320, 2, 4096, 77, 1, 4), (640, 4, 256, 77, 2, 4), (640, 2, 1024, 77, 1, 16), (320, 2, 64, 5, 1, 8)])
def test_cross_attn_block_with_lora_slabs_and_recording(ops, dev, C, B, HW, L, kv_div, r):
    """The finetuned model's form of fd_cross_attn_block: LoRA slabs on attn2.to_q / to_out (LoRAAttnProcessor.__call__, exp-1 main:811-817) and everything the
    backward consumes written as extra outputs -- n2, LayerNorm2 statistics, q (pre-scaled for d = 40, unscaled for d = 80, as fd_attn_bwd_* take it), t_q, o, the
    log-sum-exp, t_o, LayerNorm3 statistics -- against the separate launches (lora_linear_fwd + attn_fwd + layernorm) and torch fp32; the recording launch and the
    plain LoRA launch must agree bit for bit (R1 and R3 evaluate the same function)."""
    import torch.nn.functional as F
    from finetune_fair_diffusion_amd.layers import LoRAPair, lora_linear_fwd, Linear
    H, d, M, Bk = 8, C // 8, B * HW, B // kv_div
    x = rnd(M, C, dev=dev, seed=1)
    g2, b2 = rnd(C, dev=dev, dtype=torch.float32, seed=2) * 0.2 + 1, rnd(C, dev=dev, dtype=torch.float32, seed=3) * 0.2
    g3, b3 = rnd(C, dev=dev, dtype=torch.float32, seed=4) * 0.2 + 1, rnd(C, dev=dev, dtype=torch.float32, seed=5) * 0.2
    wq, wo = rnd(C, C, dev=dev, scale=C ** -0.5, seed=6), rnd(C, C, dev=dev, scale=C ** -0.5, seed=7)
    bo = rnd(C, dev=dev, dtype=torch.float32, seed=8) * 0.1
    k, v = rnd(Bk * L, C, dev=dev, seed=9), rnd(Bk * L, C, dev=dev, seed=10)
    vt = ops.transpose_btc(v, Bk, L, C, ops.CROSS_LP)
    pairs = []
    for seed in (11, 12):
        p = LoRAPair.__new__(LoRAPair)
        p.r, p.rp, p.K, p.N = r, (r + 7) // 8 * 8, C, C
        p.down16 = torch.zeros(p.rp, C, dtype=torch.float16, device=dev); p.down16[:r] = rnd(r, C, dev=dev, scale=C ** -0.5, seed=seed)
        p.up16 = torch.zeros(C, p.rp, dtype=torch.float16, device=dev); p.up16[:, :r] = rnd(C, r, dev=dev, scale=0.3, seed=seed + 10)
        pairs.append(p)
    lq, lo_ = pairs
    qs = ops.q_prescale(d)
    assert ops.cross_block_ok(M, C, H, L, HW, lq.rp)
    y, yn, st, rec = ops.cross_attn_block(x, (g2, b2, 1e-5), wq, k, vt, L, wo, bo, (g3, b3, 1e-5), H, HW, kv_div, lora_q=lq, lora_o=lo_, record=True, q_prescaled=qs is not None)
    y0, yn0, _, none = ops.cross_attn_block(x, (g2, b2, 1e-5), wq, k, vt, L, wo, bo, (g3, b3, 1e-5), H, HW, kv_div, lora_q=lq, lora_o=lo_)
    assert none is None and torch.equal(y, y0) and torch.equal(yn, yn0), "recording changes the values"
    # the separate launches
    n2s, ln2s = ops.layernorm(x, g2, b2, 1e-5, save_stats=True)
    tq = ops.gemm(n2s, lq.down16)
    q2 = ops.gemm(n2s, wq, a2=tq, b2=lq.up16, colscale=(qs, C) if qs is not None else None)
    o2, lse2 = ops.attn_fwd(q2, k, v, B, H, HW, L, d, kv_div, need_lse=True, prescaled=qs is not None)
    to = ops.gemm(o2, lo_.down16)
    ys = ops.gemm(o2, wo, a2=to, b2=lo_.up16, bias=bo, residual=x)
    yns, ln3s = ops.layernorm(ys, g3, b3, 1e-5, save_stats=True)
    # LayerNorm2 is the separate kernel's arithmetic statement for statement: the statistics agree bit for bit at C = 320; outputs (and, at wider C, the statistics) can
    # move by one ulp with the compiler's FMA contraction, as they do between layernorm_kernel's own row-count variants
    if C == 320:
        assert torch.equal(rec["ln2"], ln2s)
    check("LayerNorm2 statistics", rec["ln2"], ln2s, 1e-5)
    check("n2", rec["n2"], n2s.float(), 1e-3)
    check("t_q", rec["tq2"], tq.float(), 1e-3)
    check("q (as the backward takes it)", rec["q2"], q2.float(), 2e-3)
    check("o", rec["o2"], o2.float(), 3e-3)
    check("lse", rec["lse2"], lse2, 1e-3)
    check("t_o", rec["to2"], to.float(), 3e-3)
    check("y vs the separate launches", y, ys.float(), 2e-3)
    check("LayerNorm3(y) vs the separate launches", yn, yns.float(), 3e-3)
    check("LayerNorm3 statistics", st, ln3s, 2e-3)
    # torch fp32 on the same fp16 parameters
    xf = x.float()
    n2 = F.layer_norm(xf, (C,), g2, b2, 1e-5)
    q = n2 @ wq.float().t() + (n2 @ lq.down16.float().t()) @ lq.up16.float().t()
    o, lse = _attn_ref(q.view(B, HW, C), k.float().view(Bk, L, C), v.float().view(Bk, L, C), H, kv_div)
    o = o.reshape(M, C)
    yr = o @ wo.float().t() + (o @ lo_.down16.float().t()) @ lo_.up16.float().t() + bo + xf
    check("y vs fp32", y, yr, 4e-3)
    check("o vs fp32", rec["o2"], o, 4e-3)
    check("lse vs fp32", rec["lse2"], lse, 2e-3)
    # the attention backward fed with what the FUSED forward recorded (at head dim 80 the stored q is the unscaled accumulator's rounding while the
    # kernel's own softmax -- and hence lse -- ran on the pre-scaled rounding: include/fairdiff_hip.h, fd_cross_block_desc) against autograd of the fp32 statement
    qf, kf, vf = q.detach().view(B, HW, C).requires_grad_(True), k.float().view(Bk, L, C).requires_grad_(True), v.float().view(Bk, L, C).requires_grad_(True)
    oa, _ = _attn_ref(qf, kf, vf, H, kv_div)
    do = rnd(M, C, dev=dev, seed=21)
    oa.backward(do.float().view(B, HW, C))
    dko, dvo = torch.empty(Bk * L, C, dtype=torch.float32, device=dev), torch.empty(Bk * L, C, dtype=torch.float32, device=dev)
    dq, _, _ = ops.attn_bwd(rec["q2"], k, v, rec["o2"], do, rec["lse2"], B, H, HW, L, d, kv_div, dk_out=dko, dv_out=dvo, prescaled=qs is not None)
    check("dq from the fused recording vs fp32 autograd", dq, qf.grad.reshape(M, C), 5e-3)
    check("dK from the fused recording vs fp32 autograd", dko, kf.grad.reshape(Bk * L, C), 5e-3)
    check("dV from the fused recording vs fp32 autograd", dvo, vf.grad.reshape(Bk * L, C), 5e-3)


@pytest.mark.parametrize("B,H,T,d", [(2, 8, 1024, 40), (2, 8, 256, 80), (1, 4, 64, 160)])
def test_attention_strided_qkv_slices(ops, dev, B, H, T, d):
    """q, k, v as column slices of ONE [M, 3C] projection buffer and dq, dk, dv written as slices of one [M, 3C] gradient buffer (the
    fused self-attention projections): bit-identical to the contiguous call, forward and backward."""
    C = H * d
    qkv = rnd(B * T, 3 * C, dev=dev, seed=1)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    qc, kc, vc = q.contiguous(), k.contiguous(), v.contiguous()
    o_ref, lse_ref = ops.attn_fwd(qc, kc, vc, B, H, T, T, d, 1, need_lse=True)
    o, lse = ops.attn_fwd(q, k, v, B, H, T, T, d, 1, need_lse=True)          # v itself, row stride 3C
    assert torch.equal(o, o_ref) and torch.equal(lse, lse_ref)
    do = rnd(B * T, C, dev=dev, seed=4)
    dq_r, dk_r, dv_r = ops.attn_bwd(qc, kc, vc, o_ref, do, lse_ref, B, H, T, T, d, 1)
    dqkv = torch.full((B * T, 3 * C), float("nan"), dtype=qkv.dtype, device=dev)
    dq, dk, dv = ops.attn_bwd(q, k, v, o, do, lse, B, H, T, T, d, 1, dqkv=dqkv)
    assert dq.data_ptr() == dqkv.data_ptr() and torch.isfinite(dqkv.float()).all()
    assert torch.equal(dqkv[:, :C], dq_r) and torch.equal(dqkv[:, C:2 * C], dk_r) and torch.equal(dqkv[:, 2 * C:], dv_r)


@pytest.mark.parametrize("M,N,R", [(4096, 320, 4), (1000, 1280, 50), (777, 768, 16)])
def test_lora_wgrad(ops, dev, M, N, R):
    RP = (R + 7) // 8 * 8
    RP = 8 if R <= 8 else 16 if R <= 16 else 32 if R <= 32 else 64
    X = rnd(M, N, dev=dev, seed=1)
    T = torch.zeros(M, RP, dtype=torch.float16, device=dev)
    T[:, :R] = rnd(M, R, dev=dev, seed=2)
    G = torch.ones(N, R, dtype=torch.float32, device=dev)
    ops.lora_wgrad(X, T, G, R, 1, R, scale=0.5)
    check("lora wgrad [N,R]", G, 1 + 0.5 * X.float().t() @ T[:, :R].float(), 1e-3)
    G2 = torch.zeros(R, N, dtype=torch.float32, device=dev)
    ops.lora_wgrad(X, T, G2, 1, N, R)
    check("lora wgrad [R,N]", G2, T[:, :R].float().t() @ X.float(), 1e-3)


@pytest.mark.parametrize("R", [4, 12])
def test_lora_wgrad_batched_equals_single_calls(ops, dev, R):
    """fd_lora_wgrad_multi (the 16 weight gradients of a transformer block in one partial + one final launch) against the single-problem
    path and fp32 torch: mixed M / N, strided X (column slices of a wider buffer) and both output layouts ([N,R] and [R,N])."""
    RP = 8 if R <= 8 else 16
    probs = []
    g = 0
    for (M, N, sl) in [(4096, 320, False), (4096, 320, True), (1000, 1280, False), (65536, 320, False), (26, 768, False), (2048, 640, True)] * 3:
        g += 1
        Xw = rnd(M, N * (3 if sl else 1), dev=dev, seed=10 + g)
        X = Xw[:, N:2 * N] if sl else Xw
        T = torch.zeros(M, RP, dtype=torch.float16, device=dev)
        T[:, :R] = rnd(M, R, dev=dev, seed=50 + g)
        probs.append((X, T, (g % 2 == 0)))
    ref, single, batched = [], [], []
    for X, T, trans in probs:
        shape = (R, X.shape[1]) if trans else (X.shape[1], R)
        r = 0.5 * (T[:, :R].float().t() @ X.float() if trans else X.float().t() @ T[:, :R].float()) + 1
        ref.append(r)
        single.append(torch.ones(shape, dtype=torch.float32, device=dev))
        batched.append(torch.ones(shape, dtype=torch.float32, device=dev))
    for (X, T, trans), G in zip(probs, single):
        ops.lora_wgrad(X, T, G, *((1, X.shape[1]) if trans else (R, 1)), R, scale=0.5)
    with ops.wgrad_batch():
        for (X, T, trans), G in zip(probs, batched):
            ops.lora_wgrad(X, T, G, *((1, X.shape[1]) if trans else (R, 1)), R, scale=0.5)
        assert float(batched[0].sum()) == batched[0].numel()        # nothing ran yet: queued until the context closes
    for i, (a, b, r) in enumerate(zip(single, batched, ref)):
        check(f"wgrad single [{i}]", a, r, 1e-3)
        check(f"wgrad batched [{i}]", b, r, 1e-3)


def test_cfg_dpm_and_adamw(ops, dev):
    n = 2 * 4 * 64
    eps = rnd(2 * n, dev=dev, dtype=torch.float32, seed=1)
    lat = rnd(n, dev=dev, dtype=torch.float32, seed=2)
    x0p = rnd(n, dev=dev, dtype=torch.float32, seed=3)
    x0o = torch.empty_like(lat)
    l0 = lat.clone()
    ops.cfg_dpm_step(eps, 7.5, lat, x0p, x0o, 0.9, 0.43, 0.8, -0.3, 0.11)
    e = eps[:n] + 7.5 * (eps[n:] - eps[:n])
    x0 = (l0 - 0.43 * e) / 0.9
    check("dpm x0", x0o, x0, 1e-5)
    check("dpm lat", lat, 0.8 * l0 + 0.3 * x0 - 0.11 * (x0 - x0p), 1e-5)
    p = rnd(5000, dev=dev, dtype=torch.float32, seed=4)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=5e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    m, v, ema = torch.zeros_like(p), torch.zeros_like(p), p.clone()
    for step in range(1, 4):
        g = rnd(5000, dev=dev, dtype=torch.float32, seed=10 + step)
        pr.grad = g.clone()
        opt.step()
        ops.adamw_ema(p, g, m, v, ema, 5e-3, 0.9, 0.999, 1e-8, 1e-2, step, 0.25)
    check("adamw", p, pr.detach(), 1e-5)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    g = rnd(1000, dev=dev, dtype=torch.float32, seed=20)
    g0 = g.clone()
    ops.grad_finite_scale(g, 0.5, flag)
    assert int(flag.item()) == 0
    check("grad scale", g, 0.5 * g0, 1e-6)
    g[17] = float("inf")
    ops.grad_finite_scale(g, 1.0, flag)
    assert int(flag.item()) == 1


def test_small_convs_and_classifier_pieces(ops, dev):
    B, H, Cin, Cout = 2, 16, 4, 64
    x = rnd(B, Cin, H, H, dev=dev, dtype=torch.float32, seed=1)
    w = rnd(Cout, Cin, 3, 3, dev=dev, dtype=torch.float32, scale=0.2, seed=2)
    bias = rnd(Cout, dev=dev, dtype=torch.float32, seed=3)
    wk = w.permute(2, 3, 1, 0).reshape(9 * Cin, Cout).contiguous()
    for stride in (1, 2):
        y, Ho, Wo = ops.conv_small_cin(x, wk, bias, B, H, H, Cin, Cout, 3, stride)
        xr = x.clone().requires_grad_(True)
        ref = F.conv2d(xr, w, bias, stride=stride, padding=1)
        check(f"small conv s{stride}", _nchw(y, B, Ho, Wo), ref, 2e-3)
        g = rnd(*ref.shape, dev=dev, seed=4)
        ref.backward(g.float())
        dx = ops.conv_small_cin_bwd(_nhwc(g), wk, B, H, H, Cin, Cout, 3, stride)
        check(f"small conv bwd s{stride}", dx, xr.grad, 2e-3)
    # the register-tiled fast path (k = 3, Cin in {3, 4}, Cout % 8 == 0) on the hot-path shapes: U-Net conv_in 4 -> 320 (fp16 NCHW latents),
    # the classifier stem 3 -> 16, stride 2, hardswish, on an odd-sized image (partial quads, clipped rows and columns), and NHWC input
    for (ci, co, hh, ww, st, act, nchw) in [(4, 320, 64, 64, 1, "none", True), (3, 16, 45, 37, 2, "hardswish", True), (4, 512, 18, 22, 1, "none", False)]:
        xs = rnd(3, ci, hh, ww, dev=dev, seed=21)
        ws = rnd(co, ci, 3, 3, dev=dev, dtype=torch.float32, scale=0.2, seed=22)
        bs = rnd(co, dev=dev, dtype=torch.float32, seed=23)
        wks = ws.permute(2, 3, 1, 0).reshape(9 * ci, co).contiguous()
        ref = F.conv2d(xs.float(), ws, bs, stride=st, padding=1)
        ref = F.hardswish(ref) if act == "hardswish" else ref
        xin = xs if nchw else xs.permute(0, 2, 3, 1).contiguous()
        y, Ho, Wo = ops.conv_small_cin(xin, wks, bs, 3, hh, ww, ci, co, 3, st, nchw=nchw, act=act)
        assert (Ho, Wo) == tuple(ref.shape[2:])
        check(f"small conv fast path {ci}->{co} {hh}x{ww} s{st} {act} {'nchw' if nchw else 'nhwc'}", _nchw(y, 3, Ho, Wo), ref, 2e-3)
    w1 = rnd(4, 4, 1, 1, dev=dev, dtype=torch.float32, seed=5)
    y, _, _ = ops.conv_small_cin(x.half(), w1.permute(2, 3, 1, 0).reshape(4, 4).contiguous(), None, B, H, H, 4, 4, 1)
    check("1x1 conv", _nchw(y, B, H, H), F.conv2d(x.half().float(), w1), 2e-3)
    # depthwise
    C = 72
    for k, s in [(3, 1), (3, 2), (5, 1), (5, 2)]:
        xd = rnd(B, C, 14, 14, dev=dev, seed=6)
        wd = rnd(C, 1, k, k, dev=dev, dtype=torch.float32, scale=0.3, seed=7)
        bd = rnd(C, dev=dev, dtype=torch.float32, seed=8)
        xr = xd.float().requires_grad_(True)
        ref = F.hardswish(F.conv2d(xr, wd, bd, stride=s, padding=(k - 1) // 2, groups=C))
        wkk = wd.reshape(C, k * k).t().contiguous()
        y, Ho, Wo = ops.dwconv(_nhwc(xd), wkk, bd, B, 14, 14, C, k, s, "hardswish")
        check(f"dwconv k{k}s{s}", _nchw(y, B, Ho, Wo), ref, 2e-3)
        lin = F.conv2d(xr, wd, None, stride=s, padding=(k - 1) // 2, groups=C)
        g = rnd(*lin.shape, dev=dev, seed=9)
        xr.grad = None
        lin.backward(g.float())
        check(f"dwconv bwd k{k}s{s}", _nchw(ops.dwconv_bwd(_nhwc(g), wkk, B, 14, 14, C, k, s), B, 14, 14), xr.grad, 3e-3)
    xa = rnd(B, 49, 120, dev=dev, seed=10)
    check("avgpool", ops.avgpool_hw(xa.reshape(-1, 120), B, 49, 120), xa.float().mean(1), 2e-3)
    s = rnd(B, 120, dev=dev, seed=11)
    check("scale ch", ops.scale_channels(xa.reshape(-1, 120), s, B, 49, 120).reshape(B, 49, 120), xa.float() * s.float()[:, None], 2e-3)
    dy = rnd(B, 49, 120, dev=dev, seed=12)
    dx, ds = ops.scale_channels_bwd(xa.reshape(-1, 120), s, dy.reshape(-1, 120), B, 49, 120)
    check("scale ch dx", dx.reshape(B, 49, 120), dy.float() * s.float()[:, None], 2e-3)
    check("scale ch ds", ds, (dy.float() * xa.float()).sum(1), 3e-3)
    check("avgpool bwd", ops.avgpool_hw_bwd(s, B, 49, 120).reshape(B, 49, 120), (s.float() / 49)[:, None].expand(B, 49, 120), 2e-3)
    # crop + resize, incl. a box leaving the image
    img = rnd(B, 3, 64, 64, dev=dev, seed=13)
    boxes = torch.tensor([[8, 8, 56, 56], [-6, 10, 40, 70]], dtype=torch.int32, device=dev)
    chips = ops.crop_resize(img, boxes, -1.0, 28)
    for i, bb in enumerate(boxes.tolist()):
        im = img[i].float().requires_grad_(True)
        l, r, bt, tp = max(bb[0], 0), min(bb[2], 64), max(bb[1], 0), min(bb[3], 64)
        face = F.pad(im[:, bt:tp, l:r], [max(-bb[0], 0), max(bb[2] - 64, 0), max(-bb[1], 0), max(bb[3] - 64, 0)], value=-1.0)
        ref = F.interpolate(face[None], size=[28, 28], mode="bilinear", align_corners=False)[0]
        check(f"crop_resize {i}", chips[i], ref, 2e-3)
        g = rnd(3, 28, 28, dev=dev, dtype=torch.float32, seed=14 + i)
        ref.backward(g)
        gfull = torch.zeros(B, 3, 28, 28, device=dev)
        gfull[i] = g
        dimg = ops.crop_resize_bwd(gfull, boxes, B, 64, 64, 28)
        check(f"crop_resize bwd {i}", dimg[i], im.grad, 1e-4)
    pre = rnd(B * 16, 4, dev=dev, seed=20) * 2
    y = ops.nhwc_to_nchw(pre, B, 16, 3, out_dtype=torch.float16, lo=-1.0, hi=1.0)
    check("nhwc_to_nchw clamp", y, pre[:, :3].float().reshape(B, 16, 3).permute(0, 2, 1).clamp(-1, 1), 1e-3)
    dimg = rnd(B, 3, 16, dev=dev, dtype=torch.float32, seed=21)
    pm = pre[:, :3].float().reshape(B, 16, 3).permute(0, 2, 1)
    check("clamp bwd", ops.clamp_bwd(pre, dimg, B, 16, 3), dimg * ((pm >= -1) & (pm <= 1)), 1e-6)


@pytest.mark.parametrize("M,F,K", [(300, 64, 64), (4100, 1280, 320), (65536, 1280, 320), (1024, 5120, 1280)])
def test_gemm_fused_geglu_bit_identical(ops, dev, M, F, K):
    """FF1 with the GEGLU gate fused into the epilogue == projection GEMM followed by fd_geglu_fwd, bit for bit
    (both halves are rounded to fp16 before the gate in either path)."""
    a = rnd(M, K, dev=dev, seed=1)
    w = rnd(2 * F, K, dev=dev, scale=0.1, seed=2)
    bias = rnd(2 * F, dev=dev, dtype=torch.float32, seed=3)
    ref = ops.geglu(ops.gemm(a, w, bias=bias))
    wi, bi = ops.interleave_geglu(w, bias)
    got = ops.gemm(a, wi, bias=bi, act="geglu")
    assert got.shape == (M, F) and torch.equal(got, ref)
    x = a.float() @ w.float().t() + bias
    check(f"geglu {M}x{F}x{K}", got, x[:, :F] * torch.nn.functional.gelu(x[:, F:]), 5e-3)


def test_gemm_fused_geglu_with_pregate_output_and_interleaved_backward(ops, dev):
    """Recording forwards: the fused FF1 also keeps the pre-gate projection (interleaved columns); its backward kernel equals
    fd_geglu_bwd on the de-interleaved tensors."""
    M, F, K = 4100, 1280, 320
    a = rnd(M, K, dev=dev, seed=1)
    w = rnd(2 * F, K, dev=dev, scale=0.1, seed=2)
    bias = rnd(2 * F, dev=dev, dtype=torch.float32, seed=3)
    proj = ops.gemm(a, w, bias=bias)
    wi, bi = ops.interleave_geglu(w, bias)
    aux = torch.empty(M, 2 * F, dtype=torch.float16, device=dev)
    gg = ops.gemm(a, wi, bias=bi, act="geglu", aux=aux)
    assert torch.equal(gg, ops.geglu(proj))
    assert torch.equal(aux[:, 0::2], proj[:, :F]) and torch.equal(aux[:, 1::2], proj[:, F:])
    dy = rnd(M, F, dev=dev, seed=4)
    d_il = ops.geglu_bwd_interleaved(aux, dy)
    d_ref = ops.geglu_bwd(proj, dy)
    assert torch.equal(d_il[:, 0::2], d_ref[:, :F]) and torch.equal(d_il[:, 1::2], d_ref[:, F:])


@pytest.mark.parametrize("N,K,S", [(1, 8, 3), (5, 8, 20), (16, 8, 100), (64, 16, 100), (130, 16, 12), (64, 2, 10)])
def test_ot_assign_device_solver_is_exact(ops, dev, N, K, S):
    """fd_ot_assign_sum (exp-3 :1488-1536 ``ot.emd(ones(N), counts, M)`` per Monte-Carlo draw) against scipy's exact assignment on the
    capacity-replicated matrix: every draw respects the capacities and reaches the optimal cost (1e-12 relative, fp64); with continuous
    random costs the optimum is unique, so the summed plans are EQUAL.  The degenerate all-equal cost matrix must still give a feasible
    optimum."""
    import numpy as np
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(1000 * N + K)
    for degenerate in (False, True):
        M = np.full((N, K), 0.75) if degenerate else np.sqrt(rng.random((N, K)) * 2.0)
        counts = np.stack([np.bincount(rng.integers(0, K, N), minlength=K) for _ in range(S)]).astype(np.int32)
        plan, seats = ops.ot_assign_sum(torch.from_numpy(M).to(dev), torch.from_numpy(counts).to(dev), seats=True)
        torch.cuda.synchronize()
        plan, seats = plan.cpu().numpy(), seats.cpu().numpy()
        ref = np.zeros((N, K))
        for s in range(S):
            assert (np.bincount(seats[s], minlength=K) == counts[s]).all(), f"draw {s}: capacities violated"
            cols = np.repeat(np.arange(K), counts[s])
            r, c = linear_sum_assignment(M[:, cols])
            opt = M[r, cols[c]].sum()
            got = M[np.arange(N), seats[s]].sum()
            assert abs(got - opt) <= 1e-12 * max(1.0, abs(opt)), f"draw {s}: cost {got} vs optimal {opt}"
            ref[r, cols[c]] += 1.0
        onehot = np.zeros((N, K))
        for s in range(S):
            onehot[np.arange(N), seats[s]] += 1.0
        assert (plan == onehot).all(), "summed plan is not the sum of the per-draw seats"
        assert plan.sum() == N * S
        if not degenerate:
            assert (plan == ref).all(), f"summed plan differs from the host solver in {(plan != ref).sum()} entries"
    # accumulation into an existing plan
    p2 = ops.ot_assign_sum(torch.from_numpy(M).to(dev), torch.from_numpy(counts).to(dev), plan=torch.from_numpy(plan).to(dev).float())
    assert float(p2.sum()) == 2 * N * S
