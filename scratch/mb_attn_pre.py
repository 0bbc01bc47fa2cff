"""Isolated A/B of the d = 40 attention kernels with and without "pre-scaled q" (negative scale of the C-ABI), and of the QKV projection with colscale."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n
def case(B, H, T, Tk=None, kv_div=1, d=40):
    Tk = Tk or T; C = H * d; Bk = B // kv_div
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B * T, C, generator=g).to(dev).half(); k = torch.randn(Bk * Tk, C, generator=g).to(dev).half(); v = torch.randn(Bk * Tk, C, generator=g).to(dev).half()
    do = torch.randn(B * T, C, generator=g).to(dev).half()
    fac = ops.q_prescale(d); qp = (q.float() * fac).half()
    for rep in range(2):
        out = []
        for pre, qq in ((False, q), (True, qp)):
            o, lse = ops.attn_fwd(qq, k, None, B, H, T, Tk, d, kv_div, need_lse=True, v=v, prescaled=pre)
            dko = torch.empty(Bk * Tk, C, dtype=torch.float32, device=dev); dvo = torch.empty_like(dko)
            f = timeit(lambda: ops.attn_fwd(qq, k, None, B, H, T, Tk, d, kv_div, need_lse=True, v=v, prescaled=pre))
            bw = timeit(lambda: ops.attn_bwd(qq, k, v, o, do, lse, B, H, T, Tk, d, kv_div, dk_out=dko, dv_out=dvo, prescaled=pre))
            out.append((f, bw))
        print(f"B{B} H{H} T{T}x{Tk} d{d} kv_div{kv_div}: fwd {out[0][0]:7.1f} -> {out[1][0]:7.1f} us   bwd (dq + dkdv + slab sums) {out[0][1]:7.1f} -> {out[1][1]:7.1f} us   (multiply-add per score -> pre-scaled q)", flush=True)
case(16, 8, 4096)
case(8, 8, 4096)
case(16, 8, 4096, Tk=77, kv_div=8)
g = torch.Generator().manual_seed(1)
for (M, N, K) in [(65536, 960, 320), (65536, 320, 320)]:
    a = torch.randn(M, K, generator=g).to(dev).half(); b = (torch.randn(N, K, generator=g) * 0.1).to(dev).half()
    t0 = timeit(lambda: ops.gemm(a, b), 40); t1 = timeit(lambda: ops.gemm(a, b, colscale=(0.228, 320)), 40)
    t0 = timeit(lambda: ops.gemm(a, b), 40); t1 = timeit(lambda: ops.gemm(a, b, colscale=(0.228, 320)), 40)
    print(f"gemm {M}x{N}x{K}: {t0:.1f} us, with colscale on the first 320 columns {t1:.1f} us")
