"""Python wrappers over the C-ABI (include/fairdiff_hip.h): torch tensors in, raw device
pointers + the current HIP stream out.  torch is plumbing here (device memory, streams);
every op fails loudly if the HIP library is missing or a kernel reports an error.
"""
import ctypes
import os
import math

import torch

from . import lib as _lib

ACT = dict(none=0, silu=1, quick_gelu=2, gelu=3, relu=4, hardswish=5, hardsigmoid=6, geglu=7)
CONV_NORMAL, CONV_STRIDE2, CONV_UP2, CONV_TRANS2 = 0, 1, 2, 3
F16, F32 = _lib.torch_working_dtype(), torch.float32      # F16 = the 16-bit WORKING dtype (fp16, or bf16 under FD_DTYPE=bf16)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, "fairdiff ops need device tensors (no CPU path)"
    return ctypes.c_void_p(t.data_ptr())


# Race detector (tests/test_fullsize_gpu.py, scratch/diag_hazard.py): DELAY = (probability, max_cycles, random.Random) makes a call spin the
# CURRENT stream for a random number of cycles first (torch.cuda._sleep), which shifts the relative timing of the step's streams; results must not change.
DELAY = None


def _call(name, *args):
    if DELAY is not None and DELAY[2].random() < DELAY[0]:
        torch.cuda._sleep(int(DELAY[2].random() * DELAY[1]))
    L = _lib.get()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {L.fd_last_error().decode()}")


_gemm_ws = {}


def gemm_workspace():
    """64 MiB fp32 split-K workspace per (device, stream) -- kernels of two streams may run concurrently (step.py runs the R1 and R2
    rollouts on two streams) and must not share it (owned by the caller of the C-ABI, as every buffer is)."""
    dev = torch.cuda.current_device()
    key = (dev, torch.cuda.current_stream().cuda_stream)
    t = _gemm_ws.get(key)
    if t is None:
        t = torch.empty(16 << 20, dtype=F32, device=torch.device("cuda", dev))
        _gemm_ws[key] = t
    return t


class OpTimer:
    """Per-launch HIP-event timing of the MFMA GEMM/conv kernel family on the stream they are launched on
    (torch's current stream); used by bench.py's roofline pass, never inside the timed region."""

    def __init__(self):
        self.shapes = []   # (M, N, K, K2, conv_mode or -1, act, has_residual, split) per record
        self.records = []  # (rocprof kernel name, flops, algorithmic bytes, start_event, end_event, is_split_k)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for variant, flops, nbytes, a, b, split in self.records:
            ms = a.elapsed_time(b)
            e = agg.setdefault(variant, [0, 0.0, 0.0, 0.0, 0])
            e[0] += 1; e[1] += flops; e[2] += ms; e[3] += nbytes; e[4] += int(split)
        self.by_shape = {}
        for (variant, flops, nbytes, a, b, split), shp in zip(self.records, self.shapes):
            e = self.by_shape.setdefault((variant,) + shp, [0, 0.0, 0.0])
            e[0] += 1; e[1] += flops; e[2] += a.elapsed_time(b)
        return {k: dict(launches=v[0], flops=v[1], ms=v[2], bytes=v[3], splitk_launches=v[4]) for k, v in agg.items()}

    def shape_table(self):
        """[(kernel, M, N, K, K2, conv_mode, act, residual, split, launches, total ms, avg us, TFLOP/s)] sorted by total time."""
        rows = [k + (v[0], v[2], 1e3 * v[2] / v[0], v[1] / (v[2] * 1e-3) / 1e12) for k, v in self.by_shape.items()]
        return sorted(rows, key=lambda r: -r[10])


TIMER = None


GN_STATS = os.environ.get("FD_NO_GN_STATS") is None      # A/B switch: GroupNorm statistics from the producer's epilogue (fd_gemm_desc.gn_stats)

def _gemm_call(d, conv, out=None, gn_stats=False):
    ws = gemm_workspace()
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    if gn_stats and GN_STATS:
        # the kernel fd_gemm picks decides the chunk height (its wave-tile rows); 0 = no statistics epilogue for this problem.  The buffer rides on
        # the output tensor OBJECT: ``groupnorm`` finds it there, and anything that makes a new tensor of the output (cat, slicing) drops it
        rows = _lib.get().fd_gemm_stats_rows(ctypes.byref(d))
        if rows > 0:
            nph = 4 if (conv and d.conv_mode == CONV_UP2PI) else 1          # four phase problems per launch: phase-major chunk order
            st = torch.empty((nph * ((d.M + rows - 1) // rows), d.N // 10, 2), dtype=F32, device=out.device)
            d.gn_stats = st.data_ptr()
            out.gn_stats = (st, rows)
    if TIMER is None:
        _call("fd_gemm", ctypes.byref(d), _stream())
        return
    buf = ctypes.create_string_buffer(128)
    split = _lib.get().fd_gemm_kernel_name(ctypes.byref(d), buf, 128)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _call("fd_gemm", ctypes.byref(d), _stream())
    b.record()
    nph = 4 if (conv and d.conv_mode in (4, 6)) else 1     # FD_CONV_UP2P / FD_CONV_UP2PI: four phase problems per launch
    flops = 2.0 * nph * d.M * d.N * (d.K + d.K2) * max(d.batch, 1)
    # algorithmic bytes: every operand element read once, the output written once (3x3 gather: the Cin-wide input rows, not 9x)
    a_elems = (d.Bn * d.H * d.W * d.Cin) if conv else d.M * d.K
    nbytes = 2.0 * max(d.batch, 1) * (a_elems + d.M * d.K2 + nph * d.N * (d.K + d.K2) + nph * d.M * d.N)
    # keyed by the rocprof kernel name: split-K and plain launches of one instantiation are ONE entry (the events bracket the
    # splitk_reduce_kernel of a split launch together with its GEMM)
    TIMER.records.append((buf.value.decode(), flops, nbytes, a, b, split > 1))
    TIMER.shapes.append((d.M, d.N, d.K, d.K2, d.conv_mode if conv else -1, int(d.act), int(bool(d.residual)), split))


def _chk(t, dtype=F16):
    assert t.dtype == dtype and t.is_contiguous(), (t.dtype, t.shape, t.stride())
    return t


# ----------------------------------------------------------------------------- GEMM / conv
def gemm(a, b, *, a2=None, b2=None, bias=None, rowbias=None, rows_per_batch=0, residual=None, act="none", alpha=1.0,
         out=None, out_dtype=F16, n=None, aux=None, gn_stats=False, ln=None, colscale=None):
    """C[M,N] = act(alpha*(a.b^T + a2.b2^T) + bias + rowbias) + residual.  a:[M,K] (row stride free), b:[N,K].
    act="geglu": b / bias rows interleaved (value_c, gate_c) -> C[M, N/2] = value * gelu(gate) (see ``interleave_geglu``)."""
    M, K = a.shape
    N = b.shape[0] if n is None else n
    assert b.shape[1] == K and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        out = torch.empty((M, N // 2 if act == "geglu" else N), dtype=out_dtype, device=a.device)
    d = _lib.GemmDesc()
    d.A, d.lda, d.B, d.ldb = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0)
    if a2 is not None:
        assert a2.shape[0] == M and a2.shape[1] == b2.shape[1] and a2.stride(1) == 1 and b2.stride(1) == 1
        d.A2, d.lda2, d.B2, d.ldb2, d.K2 = a2.data_ptr(), a2.stride(0), b2.data_ptr(), b2.stride(0), a2.shape[1]
    d.C, d.ldc = out.data_ptr(), out.stride(0)
    if bias is not None:
        d.bias = _chk(bias, F32).data_ptr()
    if rowbias is not None:
        d.rowbias, d.ld_rowbias, d.rows_per_batch = _chk(rowbias).data_ptr(), rowbias.stride(0), rows_per_batch
    if residual is not None:
        assert residual.dtype == F16 and residual.stride(1) == 1
        d.residual, d.ldr = residual.data_ptr(), residual.stride(0)
    if aux is not None:     # act="geglu": second output, the pre-gate projection [M, N] in interleaved column order
        assert act == "geglu" and residual is None and aux.shape == (M, N) and aux.dtype == F16 and aux.is_contiguous()
        d.residual, d.ldr = aux.data_ptr(), N
    d.alpha, d.M, d.N, d.K = alpha, M, N, K
    d.act, d.out_dtype, d.batch = ACT[act], 1 if out.dtype == F32 else 0, 1
    if colscale is not None:        # (factor, columns): the first ``columns`` output columns times ``factor`` in fp32 before rounding (pre-scaled q)
        d.colscale, d.colscale_cols = colscale
    if ln is None:
        _gemm_call(d, False, out, gn_stats)
        return out
    # ln = (gamma, beta, eps): also return LayerNorm(out) and its per-row statistics (fd_layernorm_fwd; the form that wrote them from the GEMM's own
    # epilogue was never a win inside the step and lives in scratch/gemm_ln_epilogue_experiment.h)
    _gemm_call(d, False, out, False)
    y, st = layernorm(out, ln[0], ln[1], ln[2], save_stats=True)
    return out, y, st


def interleave_geglu(w, bias):
    """diffusers GEGLU projects to [value | gate] halves; the fused epilogue wants (value_c, gate_c) adjacent."""
    F = w.shape[0] // 2
    wi = torch.stack([w[:F], w[F:]], dim=1).reshape(2 * F, w.shape[1]).contiguous()
    bi = torch.stack([bias[:F], bias[F:]], dim=1).reshape(2 * F).contiguous() if bias is not None else None
    return wi, bi


def bgemm(a, b, *, alpha=1.0, out=None):
    """Strided-batch C[z] = alpha * a[z] . b[z]^T ; a:[Z,M,K], b:[Z,N,K] contiguous fp16."""
    Z, M, K = a.shape
    N = b.shape[1]
    _chk(a), _chk(b)
    if out is None:
        out = torch.empty((Z, M, N), dtype=F16, device=a.device)
    d = _lib.GemmDesc()
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = a.data_ptr(), K, b.data_ptr(), K, out.data_ptr(), N
    d.alpha, d.M, d.N, d.K, d.batch = alpha, M, N, K, Z
    d.sA, d.sB, d.sC = M * K, N * K, M * N
    _gemm_call(d, False)
    return out


def gemm_batched_into(a, b, out_view, bias, residual, Z, rows):
    """out_view[z] = a[z*rows:(z+1)*rows] . b^T + bias + residual   for z < Z, where ``out_view`` [Z,rows,N] is a strided view
    (row stride N, any batch stride) into a larger buffer and ``residual`` [rows,N] is shared by all z (ViT position table)."""
    K, N = a.shape[1], b.shape[0]
    assert a.shape[0] == Z * rows and out_view.shape == (Z, rows, N) and out_view.stride(2) == 1 and a.is_contiguous()
    d = _lib.GemmDesc()
    d.A, d.lda, d.B, d.ldb = a.data_ptr(), K, b.data_ptr(), b.stride(0)
    d.C, d.ldc = out_view.data_ptr(), out_view.stride(1)
    if bias is not None:
        d.bias = _chk(bias, F32).data_ptr()
    if residual is not None:
        d.residual, d.ldr, d.sR = _chk(residual).data_ptr(), residual.stride(0), 0
    d.alpha, d.M, d.N, d.K, d.batch = 1.0, rows, N, K, Z
    d.sA, d.sB, d.sC = rows * K, 0, out_view.stride(0)
    _gemm_call(d, False)
    return out_view


def gemm_batched_from(a_view, b, Z, rows):
    """C[z*rows:(z+1)*rows] = a_view[z] . b^T with ``a_view`` [Z,rows,K] a strided view (unit column stride)."""
    K, N = a_view.shape[2], b.shape[0]
    assert a_view.shape[:2] == (Z, rows) and a_view.stride(2) == 1 and b.shape[1] == K
    out = torch.empty((Z * rows, N), dtype=F16, device=b.device)
    d = _lib.GemmDesc()
    d.A, d.lda, d.B, d.ldb = a_view.data_ptr(), a_view.stride(1), b.data_ptr(), b.stride(0)
    d.C, d.ldc = out.data_ptr(), N
    d.alpha, d.M, d.N, d.K, d.batch = 1.0, rows, N, K, Z
    d.sA, d.sB, d.sC = a_view.stride(0), 0, rows * N
    _gemm_call(d, False)
    return out


def conv3x3(x, w, B, H, W, *, mode=CONV_NORMAL, bias=None, rowbias=None, residual=None, act="none", out=None, gn_stats=False):
    """Implicit-GEMM 3x3 conv (pad 1).  x: [B*H*W, Cin] channels-last fp16, w: [Cout, 9*Cin] (ky,kx,ci order).
    Returns ([B*Ho*Wo, Cout], Ho, Wo)."""
    Cin = x.shape[1]
    Cout = w.shape[0]
    assert w.shape[1] == 9 * Cin and x.shape[0] == B * H * W
    _chk(x), _chk(w)
    if mode == CONV_NORMAL:
        Ho, Wo = H, W
    elif mode == CONV_STRIDE2:
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
    else:
        Ho, Wo = 2 * H, 2 * W
    M = B * Ho * Wo
    if out is None:
        out = torch.empty((M, Cout), dtype=F16, device=x.device)
    d = _lib.GemmDesc()
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = x.data_ptr(), Cin, w.data_ptr(), 9 * Cin, out.data_ptr(), out.stride(0)
    if bias is not None:
        d.bias = _chk(bias, F32).data_ptr()
    if rowbias is not None:
        # one row per image, or a single row shared by the whole batch (time embedding: same t for all samples)
        d.rowbias, d.ld_rowbias, d.rows_per_batch = _chk(rowbias).data_ptr(), rowbias.stride(0), (M if rowbias.shape[0] == 1 else Ho * Wo)
    if residual is not None:
        d.residual, d.ldr = _chk(residual).data_ptr(), residual.stride(0)
    d.alpha, d.M, d.N, d.K, d.act, d.batch = 1.0, M, Cout, 9 * Cin, ACT[act], 1
    d.out_dtype = 1 if out.dtype == F32 else 0
    d.conv, d.conv_mode, d.Bn, d.H, d.W, d.Cin, d.Ho, d.Wo = 1, mode, B, H, W, Cin, Ho, Wo
    _gemm_call(d, True, out, gn_stats)
    return out, Ho, Wo


CONV_UP2P, CONV_UP2P_BWD, CONV_UP2PI = 4, 5, 6
_BIG_TILES = (256320, 128320, 128160, 256128, 256256, 512128)
_NO_UP2P = os.environ.get("FD_NO_UP2P") is not None      # A/B switch: nearest-up2 convs as 3x3 gathers at the high resolution


def _up2p_desc(x, w, out, B, H, W, Cin, Cout, bwd, bias=None):
    d = _lib.GemmDesc()
    if bwd:   # x = dOut [B*2H*2W, Cin] (high-res), out [B*H*W, Cout]
        d.K, d.conv_mode, d.H, d.W = 16 * Cin, CONV_UP2P_BWD, 2 * H, 2 * W
    else:     # x [B*H*W, Cin], out = the channels-last result [B*2H*2W, Cout]: the epilogue maps the rows of the four phases
        d.K, d.conv_mode, d.H, d.W = 4 * Cin, CONV_UP2PI, H, W
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = x.data_ptr(), Cin, w.data_ptr(), d.K, out.data_ptr(), Cout
    if bias is not None:
        d.bias = _chk(bias, F32).data_ptr()
    d.alpha, d.M, d.N, d.act, d.batch, d.out_dtype = 1.0, B * H * W, Cout, ACT["none"], 1, 0
    d.conv, d.Bn, d.Cin, d.Ho, d.Wo = 1, B, Cin, H, W
    ws = gemm_workspace()
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    return d


def conv_up2(x, conv, B, H, W):
    """Upsample2D: conv3x3(nearest-up2(x)) + bias.  x [B*H*W, Cin] -> ([B*2H*2W, Cout], 2H, 2W).  Evaluated as four 2x2-tap phase problems
    over the low-res input (4/9 of the multiply-adds) when the big-tile kernels take the shape, else as a 3x3 gather at the high resolution."""
    Cin, Cout = conv.cin, conv.cout
    if not _NO_UP2P and Cin % 64 == 0 and Cout % 8 == 0:
        out = torch.empty((4 * B * H * W, Cout), dtype=F16, device=x.device)
        # the four phases go straight into the channels-last result (FD_CONV_UP2PI: the epilogue maps rows), and the
        # epilogue leaves the GroupNorm statistics of the result behind in phase-major chunk order (``per`` = chunks per image and phase)
        d = _up2p_desc(_chk(x), conv.wk_up2p, out, B, H, W, Cin, Cout, False, conv.bias)
        if _lib.get().fd_gemm_tile(ctypes.byref(d)) in _BIG_TILES:
            _gemm_call(d, True, out, gn_stats=(H * W) % 32 == 0)
            if getattr(out, "gn_stats", None) is not None:
                out.gn_stats = out.gn_stats + ((H * W) // 32,)
            return out, 2 * H, 2 * W
    return conv3x3(x, conv.wk, B, H, W, mode=CONV_UP2, bias=conv.bias)


def conv_up2_bwd(dy, conv, B, H, W):
    """Input gradient of ``conv_up2``: dy [B*2H*2W, Cout] -> [B*H*W, Cin] (H, W = low resolution)."""
    Cin, Cout = conv.cin, conv.cout
    if not _NO_UP2P and Cout % 64 == 0 and Cin % 8 == 0:
        out = torch.empty((B * H * W, Cin), dtype=F16, device=dy.device)
        d = _up2p_desc(_chk(dy), conv.wd_up2p, out, B, H, W, Cout, Cin, True)
        if _lib.get().fd_gemm_tile(ctypes.byref(d)) % 1000000 in _BIG_TILES:
            _gemm_call(d, True)
            return out
    dxu, _, _ = conv3x3(dy, conv.wd, B, 2 * H, 2 * W)          # grad at the upsampled resolution
    return downsum2x2(dxu, B, H, W, Cin)                        # nearest-upsample backward


def conv_small_cin(x, w, bias, B, H, W, Cin, Cout, k, stride=1, nchw=True, act="none"):
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty((B * Ho * Wo, Cout), dtype=F16, device=x.device)
    assert x.is_contiguous() and x.dtype in (F16, F32)
    _call("fd_conv_small_cin", _p(x), int(x.dtype == F32), int(nchw), _p(_chk(w, F32)), _p(bias), _p(y), B, H, W, Cin, Cout, k, stride,
          ACT[act], _stream())
    return y, Ho, Wo


def conv_small_cin_bwd(dy, w, B, H, W, Cin, Cout, k, stride=1, scale=1.0):
    dx = torch.empty((B, Cin, H, W), dtype=F32, device=dy.device)
    _call("fd_conv_small_cin_bwd", _p(_chk(dy)), _p(_chk(w, F32)), _p(dx), B, H, W, Cin, Cout, k, stride, scale, _stream())
    return dx


def nhwc_to_nchw(x, B, HW, C, *, out_dtype=F32, scale=1.0, lo=-math.inf, hi=math.inf):
    y = torch.empty((B, C, HW), dtype=out_dtype, device=x.device)
    _call("fd_nhwc_to_nchw", _p(x), x.stride(0), _p(y), int(out_dtype == F32), B, HW, C, scale, lo, hi, _stream())
    return y


def clamp_bwd(pre, dimg, B, HW, C, lo=-1.0, hi=1.0):
    out = torch.empty_like(dimg)
    _call("fd_clamp_bwd", _p(pre), pre.stride(0), _p(_chk(dimg, F32)), _p(out), B, HW, C, lo, hi, _stream())
    return out


# ----------------------------------------------------------------------------- norms
_scratch = {}


def scratch(nfloats, device):
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    t = _scratch.get(key)
    if t is None or t.numel() < nfloats:
        t = torch.empty(max(nfloats, 1 << 22), dtype=F32, device=device)
        _scratch[key] = t
    return t


def groupnorm(x1, x2, B, HW, groups, eps, gamma, beta, silu):
    """y = act(GroupNorm(cat(x1, x2))) over channels-last [B*HW, C]; returns (y, stats [B,groups,2] for the backward)."""
    C1, C2 = x1.shape[1], (x2.shape[1] if x2 is not None else 0)
    st = torch.empty((B, groups, 2), dtype=F32, device=x1.device)
    y = torch.empty((B * HW, C1 + C2), dtype=F16, device=x1.device)
    s1, s2 = getattr(x1, "gn_stats", None), (getattr(x2, "gn_stats", None) if x2 is not None else None)
    if (s1 is not None and (x2 is None or s2 is not None) and HW % s1[1] == 0 and (s2 is None or HW % s2[1] == 0)
            and ((C1 + C2) // groups) % 10 == 0 and s1[0].shape[0] * s1[1] == B * HW and (s2 is None or s2[0].shape[0] * s2[1] == B * HW)):
        # every producer of the input left per-chunk sums behind: no statistics pass, x is read once.  A third entry of the
        # tuple = chunks per image and phase of a phase-major table (the up-sampling convolution, FD_CONV_UP2PI)
        per1 = s1[2] if len(s1) > 2 else 0
        per2 = s2[2] if (s2 is not None and len(s2) > 2) else 0
        _call("fd_groupnorm_fwd_stats_p", _p(_chk(x1)), C1, _p(x2), C2, B, HW, groups, eps, _p(gamma), _p(beta), int(silu), _p(y), _p(st),
              _p(s1[0]), s1[1], per1, _p(s2[0]) if s2 is not None else None, s2[1] if s2 is not None else 0, per2, _stream())
        return y, st
    sc = scratch(B * 64 * groups * 2, x1.device)
    _call("fd_groupnorm_fwd", _p(_chk(x1)), C1, _p(x2), C2, B, HW, groups, eps, _p(gamma), _p(beta), int(silu), _p(y), _p(st), _p(sc), _stream())
    return y, st


def groupnorm_bwd(x1, x2, dy, B, HW, groups, stats, gamma, beta, silu, add1=None, add2=None, need_dx2=True):
    C1, C2 = x1.shape[1], (x2.shape[1] if x2 is not None else 0)
    dx1 = torch.empty_like(x1)
    dx2 = torch.empty_like(x2) if (x2 is not None and need_dx2) else None
    sc = scratch(B * 65 * groups * 2, x1.device)
    _call("fd_groupnorm_bwd", _p(x1), C1, _p(x2), C2, _p(_chk(dy)), B, HW, groups, _p(stats), _p(gamma), _p(beta), int(silu), _p(sc),
          _p(add1), _p(add2), _p(dx1), _p(dx2), _stream())
    return dx1, dx2


def geglu_bwd_interleaved(proj_il, dy):
    M, F2 = proj_il.shape
    d = torch.empty_like(proj_il)
    _call("fd_geglu_bwd_interleaved", _p(_chk(proj_il)), _p(_chk(dy)), _p(d), M, F2 // 2, _stream())
    return d


def layernorm(x, gamma, beta, eps=1e-5, save_stats=False):
    M, C = x.shape
    y = torch.empty_like(x)
    st = torch.empty((M, 2), dtype=F32, device=x.device) if save_stats else None
    _call("fd_layernorm_fwd", _p(_chk(x)), _p(gamma), _p(beta), _p(y), _p(st), M, C, eps, _stream())
    return (y, st) if save_stats else y


def layernorm_bwd(x, dy, gamma, stats, add=None):
    M, C = x.shape
    dx = torch.empty_like(x)
    _call("fd_layernorm_bwd", _p(_chk(x)), _p(_chk(dy)), _p(gamma), _p(stats), _p(add), _p(dx), M, C, _stream())
    return dx


# ----------------------------------------------------------------------------- elementwise
def geglu(proj):
    M, F2 = proj.shape
    y = torch.empty((M, F2 // 2), dtype=F16, device=proj.device)
    _call("fd_geglu_fwd", _p(_chk(proj)), _p(y), M, F2 // 2, _stream())
    return y


def geglu_bwd(proj, dy):
    M, F2 = proj.shape
    d = torch.empty_like(proj)
    _call("fd_geglu_bwd", _p(proj), _p(_chk(dy)), _p(d), M, F2 // 2, _stream())
    return d


def act_fwd(x, act):
    y = torch.empty_like(x)
    _call("fd_act_fwd", _p(_chk(x)), _p(y), x.numel(), ACT[act], _stream())
    return y


def act_bwd(z, dy, act):
    dx = torch.empty_like(z)
    _call("fd_act_bwd", _p(_chk(z)), _p(_chk(dy)), _p(dx), z.numel(), ACT[act], _stream())
    return dx


def add(a, b, sa=1.0, sb=1.0, out=None):
    if out is None:
        out = torch.empty_like(a)
    _call("fd_add", _p(_chk(a)), _p(b), _p(out), a.numel(), sa, sb, _stream())
    return out


def copy_cols(src, dst, cols):
    _call("fd_copy_cols", _p(src), src.stride(0), _p(dst), dst.stride(0), src.shape[0], cols, _stream())


def _rows(t):
    """2-D fp16 operand whose rows may be strided (a column slice of a wider buffer): returns its row stride in elements."""
    assert t.dtype == F16 and t.dim() == 2 and t.stride(1) == 1, (t.dtype, t.shape, t.stride())
    return t.stride(0)


def transpose_btc(x, B, T, C, Tp=None, out=None):
    """x [B*T, C] (rows may be strided) -> [B, C, Tp] (keys past T are zero)."""
    Tp = Tp or ((T + 7) // 8 * 8)
    y = torch.empty((B, C, Tp), dtype=F16, device=x.device) if out is None else out
    assert y.shape == (B, C, Tp) and y.dtype == F16 and y.is_contiguous()
    _call("fd_transpose_btc", _p(x), _rows(x), _p(y), B, T, C, Tp, _stream())
    return y


def downsum2x2(x, B, H, W, C):
    y = torch.empty((B * H * W, C), dtype=F16, device=x.device)
    _call("fd_downsum2x2", _p(_chk(x)), _p(y), B, H, W, C, _stream())
    return y


def softmax_rows(x, scale=1.0, mask=None, mask_t=0, mask_ht=0):
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    y = torch.empty_like(x)
    _call("fd_softmax_rows", _p(_chk(x)), _p(y), rows, cols, scale, _p(mask), mask_t, mask_ht, _stream())
    return y


def softmax_rows_bwd(p, dp, scale=1.0):
    rows, cols = p.numel() // p.shape[-1], p.shape[-1]
    ds = torch.empty_like(p)
    _call("fd_softmax_rows_bwd", _p(_chk(p)), _p(_chk(dp)), _p(ds), rows, cols, scale, _stream())
    return ds


def to_f16(x, scale=1.0):
    y = torch.empty(x.shape, dtype=F16, device=x.device)
    _call("fd_cast_f32_to_f16", _p(_chk(x, F32)), _p(y), x.numel(), scale, _stream())
    return y


def to_f32(x, scale=1.0):
    y = torch.empty(x.shape, dtype=F32, device=x.device)
    _call("fd_cast_f16_to_f32", _p(_chk(x)), _p(y), x.numel(), scale, _stream())
    return y


# ----------------------------------------------------------------------------- attention
# The attention kernels consume V (forward), K (dQ) and Q / dO (dK, dV) in the row-major layout the projections write them in, through LDS
# transpose reads (ds_read_b64_tr_b16); the round-1/2 form with transposed copies made by fd_transpose_btc left the library in round 5.
LOG2E = 1.4426950408889634
# "Pre-scaled q" (round 4): for head dims with spare contraction slots (d = 40) the projection writes q * (d^-0.5 * log2 e) -- in its fp32 epilogue, one
# rounding as before (fd_gemm_desc.colscale) -- and the three attention kernels take the QK^T accumulator as the exponent's argument, the softmax
# reference point / saved log-sum-exp riding in the spare slots (csrc/attn.hip).  FD_NO_PRESCALED_Q=1 restores the multiply-add per score element.
PRESCALED_Q = os.environ.get("FD_NO_PRESCALED_Q") is None


def q_prescale(d):
    """Factor the q projection folds into its epilogue for head dim ``d``, or None where the attention kernels have no spare contraction slots."""
    return (d ** -0.5) * LOG2E if (PRESCALED_Q and d % 16 == 8) else None


def attn_fwd(q, k, v, B, H, Tq, Tk, d, kv_div=1, scale=None, need_lse=False, kv_rows=None, prescaled=False):
    """q [B*Tq, H*d], k [Bk*Tkr, H*d] (2-D; rows may be strided: column slices of a wider buffer), v: same shape and row stride as k.
    ``kv_rows``: rows per batch item of the k buffer when it is row-padded beyond the Tk keys (ViT token buffers).
    ``prescaled``: q holds q * scale * log2(e) (``q_prescale``); the C-ABI takes that as a negative ``scale``."""
    Tkr = kv_rows or Tk
    assert _rows(v) == _rows(k) and v.shape == k.shape
    o = torch.empty((q.shape[0], H * d), dtype=F16, device=q.device)
    lse = torch.empty((B, H, Tq), dtype=F32, device=q.device) if need_lse else None
    scale = scale if scale is not None else d ** -0.5
    _call("fd_attn_fwd", _p(q), _p(k), _p(v), _p(o), _p(lse), B, H, Tq, Tk, Tkr, d, kv_div,
          -scale if prescaled else scale, _rows(q), _rows(k), _stream())
    return (o, lse) if need_lse else o


def attn_bwd(q, k, v, o, do, lse, B, H, Tq, Tk, d, kv_div=1, scale=None, dk_acc=None, dv_acc=None, kv_rows=None, dqkv=None,
             dk_out=None, dv_out=None, prescaled=False):
    """Returns (dq, dk, dv).  With ``dk_acc`` / ``dv_acc`` (fp32 [Bk*Tk, C]; mandatory when kv_div > 1, i.e. shared K/V) dk/dv are ADDED
    into those buffers with fp32 atomics -- safe for launches that run concurrently on different streams.
    With ``dk_out`` / ``dv_out`` instead (fp32 [Bk*Tk, C], WRITTEN): no atomics -- every sample of a K/V group writes its own fp32 slab and the
    kv_div slabs are summed in a fixed order (``fd_sum_slabs``); the training step gives every timestep its own pair, so the shared
    cross-attention dK / dV -- and with them every LoRA gradient -- are bit-reproducible run to run and across schedules.
    q, k, v are 2-D and may be column slices of a wider buffer; with ``dqkv`` [M, 3*H*d] (self-attention, kv_div == 1) the three
    gradients are written as its column slices (returned as views), so that the projections' input gradient is ONE GEMM over K = 3*H*d."""
    scale = scale if scale is not None else d ** -0.5
    if prescaled:           # q = q_true * scale * log2(e); dq is still the gradient w.r.t. q_true (what the projection's backward wants)
        scale = -scale
    Tkr = kv_rows or Tk
    C = H * d
    assert _rows(k) == _rows(v)
    Dd = torch.empty((B, H, Tq), dtype=F32, device=q.device)     # D = rowsum(dO*O): produced inside the dq kernel, read by dk/dv
    if dqkv is not None:
        assert kv_div == 1 and dqkv.shape == (q.shape[0], 3 * C) and dqkv.is_contiguous() and Tkr == Tk and dqkv.dtype == F16
        dq, dk, dv = dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:]
        lddkv = 3 * C
    else:
        dq = torch.empty((q.shape[0], C), dtype=F16, device=q.device)
        lddkv = C
        if dk_out is not None:      # deterministic form: per-sample slabs, reduced below
            assert dk_out.dtype == F32 and dv_out.dtype == F32 and dk_out.shape == (k.shape[0], C) and dv_out.shape == dk_out.shape and Tkr == Tk
            assert dk_out.is_contiguous() and dv_out.is_contiguous() and B % kv_div == 0
            slabs = torch.empty((2, kv_div) + tuple(dk_out.shape), dtype=F32, device=q.device)
            dk, dv = slabs[0], slabs[1]
        elif dk_acc is not None:    # shared K/V (kv_div > 1) or a cross-step accumulator: fp32 buffers updated with atomics
            dk, dv = dk_acc, dv_acc
            assert dk.dtype == F32 and dv.dtype == F32 and dk.shape == (k.shape[0], C) and Tkr == Tk
        else:
            assert kv_div == 1, "shared K/V (kv_div > 1) needs the fp32 accumulators dk_acc / dv_acc"
            mk = torch.empty if Tkr == Tk else torch.zeros
            dk, dv = mk((k.shape[0], C), dtype=F16, device=q.device), mk((k.shape[0], C), dtype=F16, device=q.device)
    _call("fd_attn_bwd_dq", _p(q), _p(k), _p(v), _p(_chk(do)), _p(lse), _p(Dd), _p(_chk(o)), _p(dq), B, H, Tq, Tk, Tkr, d,
          kv_div, scale, _rows(q), _rows(k), _rows(dq), _stream())
    _call("fd_attn_bwd_dkdv", _p(q), _p(k), _p(v), _p(do), _p(lse), _p(Dd), _p(dk), _p(dv), B, H, Tq, Tk, Tkr, d, kv_div, scale,
          _rows(q), _rows(k), lddkv, 2 if dk_out is not None else int(dk_acc is not None), _stream())
    if dk_out is not None:
        n = dk_out.numel()
        _call("fd_sum_slabs", _p(dk), _p(dk_out), kv_div, n, _stream())
        _call("fd_sum_slabs", _p(dv), _p(dv_out), kv_div, n, _stream())
        dk, dv = dk_out, dv_out
    return dq, dk, dv


# The cross-attention sub-block (LayerNorm2 -> to_q -> attention over the prompt tokens -> to_out + residual -> LayerNorm3) as one launch, for forwards
# that neither record nor carry LoRA slabs (csrc/crossattn.hip).  FD_NO_FUSED_CROSS=1: the five separate launches.
FUSED_CROSS = os.environ.get("FD_NO_FUSED_CROSS") is None
FUSED_CROSS_TRAIN = os.environ.get("FD_NO_FUSED_CROSS_TRAIN") is None      # ... also where the forward carries LoRA slabs and / or records for the backward (R1 / R3)
CROSS_LP = 80          # key padding of the transposed V the fused kernel reads


CROSS_WIDTHS = tuple(int(c) for c in os.environ.get("FD_FUSED_CROSS_C", "320,640").split(",") if c)      # measurement: which levels take the fused kernel


def cross_block_ok(M, C, heads, L, rows_per_sample, rp=0):
    return (FUSED_CROSS and rp in (0, 8, 16) and C in CROSS_WIDTHS and heads == 8 and C in (320, 640) and L <= CROSS_LP and M % 64 == 0 and rows_per_sample % 64 == 0)


def cross_attn_block(x, ln2, wq, k, vt, L, wo, bo, ln3, heads, rows_per_sample, kv_div, need_stats=False, lora_q=None, lora_o=None, record=False, q_prescaled=False):
    """x [M, C] -> (y [M, C], LayerNorm3(y) [M, C], its (mean, rstd) [M, 2] or None, rec).  ln2 / ln3 = (gamma, beta, eps); k [Bk*L, C]; vt [Bk, C, Lp].
    ``lora_q`` / ``lora_o``: the LoRAPair of attn2.to_q / to_out (both or neither).  ``record``: rec = dict(n2, ln2, q2, tq2, o2, lse2, to2) for the backward (q2
    pre-scaled by softmax_scale * log2(e) iff ``q_prescaled``), else None."""
    M, C = x.shape
    d = _lib.CrossBlockDesc()
    y, yn = torch.empty_like(x), torch.empty_like(x)
    st = torch.empty((M, 2), dtype=F32, device=x.device) if (need_stats or record) else None
    d.x, d.ln2_gamma, d.ln2_beta, d.ln2_eps = _chk(x).data_ptr(), _chk(ln2[0], F32).data_ptr(), _chk(ln2[1], F32).data_ptr(), ln2[2]
    d.wq, d.k, d.vt, d.L, d.Lp = _chk(wq).data_ptr(), _chk(k).data_ptr(), _chk(vt).data_ptr(), L, vt.shape[-1]
    assert wq.shape == (C, C) and wo.shape == (C, C) and k.shape[1] == C and vt.shape[1] == C and k.shape[0] == vt.shape[0] * L
    d.wo, d.bo = _chk(wo).data_ptr(), _chk(bo, F32).data_ptr()
    d.ln3_gamma, d.ln3_beta, d.ln3_eps = _chk(ln3[0], F32).data_ptr(), _chk(ln3[1], F32).data_ptr(), ln3[2]
    d.y, d.yn, d.yn_stats = y.data_ptr(), yn.data_ptr(), (st.data_ptr() if st is not None else None)
    d.M, d.C, d.heads, d.rows_per_sample, d.kv_div, d.scale = M, C, heads, rows_per_sample, kv_div, (C // heads) ** -0.5
    rp = 0
    if lora_q is not None:
        rp = lora_q.rp
        assert lora_o is not None and lora_o.rp == rp and lora_q.down16.shape == (rp, C) and lora_q.up16.shape == (C, rp)
        d.lora_q_down, d.ld_q_down, d.lora_q_up, d.ld_q_up = lora_q.down16.data_ptr(), _rows(lora_q.down16), lora_q.up16.data_ptr(), _rows(lora_q.up16)
        d.lora_o_down, d.ld_o_down, d.lora_o_up, d.ld_o_up = lora_o.down16.data_ptr(), _rows(lora_o.down16), lora_o.up16.data_ptr(), _rows(lora_o.up16)
        d.lora_rp = rp
    rec = None
    if record:
        dev = x.device
        rec = dict(n2=torch.empty_like(x), ln2=torch.empty((M, 2), dtype=F32, device=dev), q2=torch.empty_like(x), o2=torch.empty_like(x),
                   lse2=torch.empty((M // rows_per_sample, heads, rows_per_sample), dtype=F32, device=dev),
                   tq2=torch.empty((M, rp), dtype=F16, device=dev) if rp else None, to2=torch.empty((M, rp), dtype=F16, device=dev) if rp else None)
        d.n2_out, d.ln2_stats, d.q_out, d.o_out, d.lse_out = rec["n2"].data_ptr(), rec["ln2"].data_ptr(), rec["q2"].data_ptr(), rec["o2"].data_ptr(), rec["lse2"].data_ptr()
        if rp:
            d.tq_out, d.to_out = rec["tq2"].data_ptr(), rec["to2"].data_ptr()
        d.q_prescaled = int(bool(q_prescaled))
    _call("fd_cross_attn_block", ctypes.byref(d), _stream())
    return y, yn, st, rec


# ----------------------------------------------------------------------------- LoRA / scheduler / optimizer
_WGRAD_PENDING = None      # list of (X, T, G, sn, sr, R, scale) while a ``wgrad_batch()`` context is open
_WGRAD_BATCHING = os.environ.get("FD_NO_WGRAD_BATCH") is None     # A/B switch: 0 -> every weight gradient is its own launch pair
WGRAD_MAX = 16


def lora_wgrad(X, T, G, sn, sr, R, scale=1.0):
    """G[n*sn + r*sr] += scale * sum_m X[m,n] T[m,r].  Inside ``with wgrad_batch():`` the call is queued and executed with the other
    queued problems in one partial + one final launch when the context closes (the operands are kept alive until then)."""
    if _WGRAD_BATCHING and _WGRAD_PENDING is not None and R <= 16 and X.shape[1] % 8 == 0 and X.stride(0) % 8 == 0:
        _WGRAD_PENDING.append((X, T, G, sn, sr, R, scale))
        return
    M, N = X.shape
    sc = scratch(1 << 22, X.device)
    _call("fd_lora_wgrad", _p(X), X.stride(0), _p(T), T.stride(0), _p(G), sn, sr, M, N, R, scale, _p(sc), sc.numel(), _stream())


class wgrad_batch:
    """Collects the LoRA weight-gradient problems issued inside the block (``lora_wgrad`` calls) and runs them batched on exit:
    problems of one padded rank go, up to 16 at a time, through ``fd_lora_wgrad_multi``."""

    def __enter__(self):
        global _WGRAD_PENDING
        self.outer, _WGRAD_PENDING = _WGRAD_PENDING, []
        return self

    def __exit__(self, et, ev, tb):
        global _WGRAD_PENDING
        pend, _WGRAD_PENDING = _WGRAD_PENDING, self.outer
        if et is None:
            flush_wgrads(pend)
        return False


def flush_wgrads(pend):
    by_rp = {}
    for it in pend:
        by_rp.setdefault(8 if it[5] <= 8 else 16, []).append(it)
    for rp, items in by_rp.items():
        for i in range(0, len(items), WGRAD_MAX):
            chunk = items[i:i + WGRAD_MAX]
            arr = _lib.WgradDesc.array(len(chunk))
            for d, (X, T, G, sn, sr, R, scale) in zip(arr, chunk):
                assert X.dtype == F16 and T.dtype == F16 and G.dtype == F32 and X.stride(1) == 1 and T.stride(1) == 1
                d.X, d.ldx, d.T, d.ldt, d.G = X.data_ptr(), X.stride(0), T.data_ptr(), T.stride(0), G.data_ptr()
                d.g_stride_n, d.g_stride_r, d.M, d.N, d.R, d.scale = sn, sr, X.shape[0], X.shape[1], R, scale
            sc = scratch(1 << 22, chunk[0][0].device)
            _call("fd_lora_wgrad_multi", ctypes.byref(arr), len(chunk), _p(sc), sc.numel(), _stream())


def cfg_dpm_step(eps, guidance, lat, x0_prev, x0_out, alpha_t, sigma_t, c_x, c_d0, c_d1):
    n = lat.numel()
    assert eps.numel() == 2 * n
    _call("fd_cfg_dpm_step", _p(_chk(eps, F32)), guidance, _p(_chk(lat, F32)), _p(x0_prev), _p(_chk(x0_out, F32)), alpha_t, sigma_t, c_x, c_d0,
          c_d1, n, _stream())


def grad_finite_scale(g, scale, flag):
    _call("fd_grad_finite_scale", _p(_chk(g, F32)), g.numel(), scale, _p(flag), _stream())


def adamw_ema(p, g, m, v, ema, lr, b1, b2, eps, wd, step, ema_omd):
    _call("fd_adamw_ema", _p(p), _p(g), _p(m), _p(v), _p(ema), p.numel(), lr, b1, b2, eps, wd, step, ema_omd, _stream())


# ----------------------------------------------------------------------------- classifier pieces
def dwconv(x, w, bias, B, H, W, C, k, stride, act):
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty((B * Ho * Wo, C), dtype=F16, device=x.device)
    _call("fd_dwconv_fwd", _p(_chk(x)), _p(w), _p(bias), _p(y), B, H, W, C, k, stride, ACT[act], _stream())
    return y, Ho, Wo


def dwconv_bwd(dy, w, B, H, W, C, k, stride):
    dx = torch.empty((B * H * W, C), dtype=F16, device=dy.device)
    _call("fd_dwconv_bwd", _p(_chk(dy)), _p(w), _p(dx), B, H, W, C, k, stride, _stream())
    return dx


def avgpool_hw(x, B, HW, C):
    y = torch.empty((B, C), dtype=F16, device=x.device)
    _call("fd_avgpool_hw", _p(_chk(x)), _p(y), B, HW, C, _stream())
    return y


def avgpool_hw_bwd(dy, B, HW, C, add=None):
    dx = torch.empty((B * HW, C), dtype=F16, device=dy.device)
    _call("fd_avgpool_hw_bwd", _p(_chk(dy)), _p(add), _p(dx), B, HW, C, _stream())
    return dx


def scale_channels(x, s, B, HW, C):
    y = torch.empty_like(x)
    _call("fd_scale_channels", _p(_chk(x)), _p(_chk(s)), _p(y), B, HW, C, _stream())
    return y


def scale_channels_bwd(x, s, dy, B, HW, C):
    dx = torch.empty_like(x)
    ds = torch.empty((B, C), dtype=F16, device=x.device)
    _call("fd_scale_channels_bwd", _p(x), _p(s), _p(_chk(dy)), _p(dx), _p(ds), B, HW, C, _stream())
    return dx, ds


def crop_resize(img, boxes, fill, S):
    B, _, H, W = img.shape
    chips = torch.empty((B, 3, S, S), dtype=F16, device=img.device)
    _call("fd_crop_resize_fwd", _p(_chk(img)), _p(boxes), fill, _p(chips), B, H, W, S, _stream())
    return chips


def crop_resize_bwd(dchips, boxes, B, H, W, S):
    dimg = torch.zeros((B, 3, H, W), dtype=F32, device=dchips.device)
    _call("fd_crop_resize_bwd", _p(_chk(dchips, F32)), _p(boxes), _p(dimg), B, H, W, S, _stream())
    return dimg


def patchify(chips, mean, std, P, Kp):
    N, _, S, _ = chips.shape
    g = S // P
    out = torch.empty((N * g * g, Kp), dtype=F16, device=chips.device)
    _call("fd_patchify_fwd", _p(_chk(chips)), _p(out), (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std), N, S, P, Kp, _stream())
    return out


def patchify_bwd(dpatches, std, N, S, P, scale=1.0, out=None):
    """dchips [N,3,S,S] fp32 = (or +=, when ``out`` is given) 0.5/std_c * scale * fold(dpatches)."""
    acc = out is not None
    if out is None:
        out = torch.empty((N, 3, S, S), dtype=F32, device=dpatches.device)
    _call("fd_patchify_bwd", _p(_chk(dpatches)), _p(_chk(out, F32)), (ctypes.c_float * 3)(*std), N, S, P, dpatches.shape[1], scale, int(acc), _stream())
    return out


def warp_affine(img, src_index, A, S, fill=-1.0):
    """chips [n,3,S,S] fp16 sampled from img [B,3,H,W] fp16 through the per-chip 2x3 maps A [n,6] (output pixel -> input position)."""
    n = A.shape[0]
    _, _, H, W = img.shape
    chips = torch.empty((n, 3, S, S), dtype=F16, device=img.device)
    _call("fd_warp_affine_fwd", _p(_chk(img)), _p(src_index), _p(_chk(A, F32)), fill, _p(chips), n, H, W, S, _stream())
    return chips


def warp_affine_bwd(dchips, src_index, A, dimg, S):
    """dimg [B,3,H,W] fp32 += the adjoint of ``warp_affine`` applied to dchips [n,3,S,S] fp32 (fixed-order gather: bit-reproducible)."""
    B, _, H, W = dimg.shape
    _call("fd_warp_affine_bwd", _p(_chk(dchips, F32)), _p(src_index), _p(_chk(A, F32)), _p(_chk(dimg, F32)), A.shape[0], B, H, W, S, _stream())
    return dimg


def rect_scale(dimg, rects, factors):
    B, _, H, W = dimg.shape
    _call("fd_rect_scale", _p(_chk(dimg, F32)), _p(rects), _p(_chk(factors, F32)), B, H, W, _stream())
    return dimg


def ot_assign_sum(cost, counts, plan=None, seats=False):
    """Summed 0/1 transport plans of S capacity draws (exp-3 :1488-1536): cost [N,K] f64, counts [S,K] int32 (rows sum to N), both on the
    device.  Returns plan [N,K] f32 (accumulated into ``plan`` when given) and, with ``seats``, the cell of every face per draw [S,N]."""
    N, K = cost.shape
    S = counts.shape[0]
    assert counts.shape[1] == K
    if plan is None:
        plan = torch.zeros((N, K), dtype=F32, device=cost.device)
    st = torch.empty((S, N), dtype=torch.int32, device=cost.device) if seats else None
    _call("fd_ot_assign_sum", _p(_chk(cost, torch.float64)), _p(_chk(counts, torch.int32)), _p(_chk(plan, F32)), _p(st), N, K, S, _stream())
    return (plan, st) if seats else plan


# ----------------------------------------------------------------------------- text-encoder attention
def small_attn_fwd(q, k, v, key_valid, B, H, T, d, scale, causal=True, save_p=False):
    o = torch.empty_like(q)
    P = torch.empty((B, H, T, T), dtype=F32, device=q.device) if save_p else None
    _call("fd_small_attn_fwd", _p(_chk(q)), _p(_chk(k)), _p(_chk(v)), _p(o), _p(P), _p(key_valid), B, H, T, d, scale, int(causal), _stream())
    return (o, P) if save_p else o


def small_attn_bwd(q, k, v, P, do, B, H, T, d, scale):
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    _call("fd_small_attn_bwd", _p(q), _p(k), _p(v), _p(_chk(P, F32)), _p(_chk(do)), _p(dq), _p(dk), _p(dv), B, H, T, d, scale, _stream())
    return dq, dk, dv
