"""fd_cross_attn_block against the five launches it replaces, isolated, at the step's shapes (CFG batch 16, 77 prompt tokens).  us per call, median of 5 x 20."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda:0")
def rnd(*s, scale=1.0, dt=torch.float16): return (torch.randn(*s, device=dev) * scale).to(dt)
def timeit(fn, n=20, rep=5):
    for _ in range(3): fn()
    ts = []
    for _ in range(rep):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1000 / n)
    return sorted(ts)[len(ts) // 2]
B, L, H = 16, 77, 8
print("C HW M fused_us separate_us (ln2 q attn out ln3)")
for C, HW in ((320, 4096), (640, 1024)):
    M, d, Bk = B * HW, C // 8, 2
    x = rnd(M, C); g2, b2, g3, b3 = (rnd(C, dt=torch.float32) for _ in range(4)); wq, wo = rnd(C, C, scale=C ** -0.5), rnd(C, C, scale=C ** -0.5)
    bo = rnd(C, dt=torch.float32); k, v = rnd(Bk * L, C), rnd(Bk * L, C); vt = ops.transpose_btc(v, Bk, L, C, 80)
    qs = ops.q_prescale(d)
    f = timeit(lambda: ops.cross_attn_block(x, (g2, b2, 1e-5), wq, k, vt, L, wo, bo, (g3, b3, 1e-5), H, HW, B // Bk))
    n2 = ops.layernorm(x, g2, b2); q = ops.gemm(n2, wq, colscale=(qs, C) if qs else None); o = ops.attn_fwd(q, k, v, B, H, HW, L, d, B // Bk, prescaled=qs is not None)
    y = ops.gemm(o, wo, bias=bo, residual=x)
    parts = [timeit(lambda: ops.layernorm(x, g2, b2)), timeit(lambda: ops.gemm(n2, wq, colscale=(qs, C) if qs else None)),
             timeit(lambda: ops.attn_fwd(q, k, v, B, H, HW, L, d, B // Bk, prescaled=qs is not None)), timeit(lambda: ops.gemm(o, wo, bias=bo, residual=x)),
             timeit(lambda: ops.layernorm(y, g3, b3))]
    def sep():
        n2 = ops.layernorm(x, g2, b2); q = ops.gemm(n2, wq, colscale=(qs, C) if qs else None); o = ops.attn_fwd(q, k, v, B, H, HW, L, d, B // Bk, prescaled=qs is not None)
        y = ops.gemm(o, wo, bias=bo, residual=x); return ops.layernorm(y, g3, b3)
    s = timeit(sep)
    # the finetuned model's form: LoRA slabs (rank 4 -> 8) + recording, against lora_linear_fwd x 2 + attn_fwd + the two LayerNorms with saved statistics
    from finetune_fair_diffusion_amd.layers import LoRAPair
    prs = []
    for _ in range(2):
        p_ = LoRAPair.__new__(LoRAPair); p_.r, p_.rp, p_.K, p_.N = 4, 8, C, C
        p_.down16 = torch.zeros(8, C, dtype=torch.float16, device=dev); p_.down16[:4] = rnd(4, C, scale=C ** -0.5)
        p_.up16 = torch.zeros(C, 8, dtype=torch.float16, device=dev); p_.up16[:, :4] = rnd(C, 4, scale=0.3)
        prs.append(p_)
    fr = timeit(lambda: ops.cross_attn_block(x, (g2, b2, 1e-5), wq, k, vt, L, wo, bo, (g3, b3, 1e-5), H, HW, B // Bk, lora_q=prs[0], lora_o=prs[1], record=True, q_prescaled=qs is not None))
    def sep_rec():
        n2, s2 = ops.layernorm(x, g2, b2, save_stats=True); tq = ops.gemm(n2, prs[0].down16)
        q = ops.gemm(n2, wq, a2=tq, b2=prs[0].up16, colscale=(qs, C) if qs else None)
        o, lse = ops.attn_fwd(q, k, v, B, H, HW, L, d, B // Bk, need_lse=True, prescaled=qs is not None)
        to = ops.gemm(o, prs[1].down16); y = ops.gemm(o, wo, a2=to, b2=prs[1].up16, bias=bo, residual=x)
        return ops.layernorm(y, g3, b3, save_stats=True)
    sr = timeit(sep_rec)
    print(f"{C} {HW} {M} LoRA + recording: fused {fr:.1f} us  separate (7 launches) {sr:.1f} us", flush=True)
    gf = (4.0 * M * C * C + 4.0 * M * L * C) / 1e9
    print(f"{C} {HW} {M} fused {f:.1f} us ({gf / f * 1e3:.0f} TF)  separate {s:.1f} us  parts " + " ".join(f"{p:.1f}" for p in parts), flush=True)
