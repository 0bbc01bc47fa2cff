import sys; sys.path.insert(0, "/root/repo")
import torch
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda")
def bench(name, fn, flops, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print(f"{name:40s} {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TFLOP/s")
B, H = 16, 8
for (T, d) in [(4096, 40), (1024, 80), (256, 160)]:
    C = H * d
    q, k, v = (torch.randn(B * T, C, device=dev).half() for _ in range(3))
    vt = ops.transpose_btc(v, B, T, C)
    fl = 4.0 * B * H * T * T * d
    o, lse = ops.attn_fwd(q, k, vt, B, H, T, T, d, 1, need_lse=True)
    bench(f"attn fwd T={T} d={d}", lambda: ops.attn_fwd(q, k, vt, B, H, T, T, d, 1, need_lse=True), fl)
    do = torch.randn_like(o)
    bench(f"attn bwd T={T} d={d} (prep+dq+dkdv+3 transposes)", lambda: ops.attn_bwd(q, k, v, o, do, lse, B, H, T, T, d, 1), 2.5 * fl)
