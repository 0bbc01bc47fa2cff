"""Isolated A/B of the attention kernels: transposed-copy form (fd_transpose_btc + column tiles) vs the LDS transpose-read form
(ds_read_b64_tr_b16 on the row-major tiles).  B16 H8 T4096 d40 is VERDICT r2's reference point (dq + dkdv 2.19 ms)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


def case(B, H, T, d, Tk=None, kv_div=1):
    Tk = Tk or T
    C = H * d
    g = torch.Generator().manual_seed(0)
    Bk = B // kv_div
    q = torch.randn(B * T, C, generator=g).to(dev).half()
    k = torch.randn(Bk * Tk, C, generator=g).to(dev).half()
    v = torch.randn(Bk * Tk, C, generator=g).to(dev).half()
    do = torch.randn(B * T, C, generator=g).to(dev).half()
    vt = ops.transpose_btc(v, Bk, Tk, C, (Tk + 7) // 8 * 8)
    o, lse = ops.attn_fwd(q, k, vt, B, H, T, Tk, d, kv_div, need_lse=True)
    acc = (torch.zeros(Bk * Tk, C, device=dev), torch.zeros(Bk * Tk, C, device=dev)) if kv_div > 1 else (None, None)
    fl = 4.0 * B * H * T * Tk * d
    t_tr = timeit(lambda: ops.transpose_btc(v, Bk, Tk, C, (Tk + 7) // 8 * 8))
    f0 = timeit(lambda: ops.attn_fwd(q, k, vt, B, H, T, Tk, d, kv_div, need_lse=True))
    f1 = timeit(lambda: ops.attn_fwd(q, k, None, B, H, T, Tk, d, kv_div, need_lse=True, v=v))
    b0 = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, B, H, T, Tk, d, kv_div, dk_acc=acc[0], dv_acc=acc[1], tr=False)) if T % 8 == 0 else float("nan")
    b1 = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, B, H, T, Tk, d, kv_div, dk_acc=acc[0], dv_acc=acc[1], tr=True))
    print(f"B{B} H{H} T{T}x{Tk} d{d} kv_div{kv_div}: fwd copies {f0:7.1f} us (+{t_tr:5.1f} transpose)  tr {f1:7.1f} us ({fl / f1 / 1e6:5.0f} TF/s)"
          f"   bwd copies (3 transposes incl.) {b0:7.1f} us  tr {b1:7.1f} us ({2.5 * fl / b1 / 1e6:5.0f} TF/s)", flush=True)


if __name__ == "__main__":
    case(16, 8, 4096, 40)
    case(8, 8, 4096, 40)
    case(16, 8, 4096, 40, Tk=77, kv_div=8)
    case(16, 8, 1024, 80)
    case(16, 8, 256, 160)
    case(16, 16, 257, 80)
    case(16, 12, 257, 64)
