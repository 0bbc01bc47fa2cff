import sys; sys.path.insert(0, "/root/repo")
import torch
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda")
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for mb in (84, 336, 1344):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device=dev).half(); y = torch.empty_like(x)
    t = bench(lambda: y.zero_()); print(f"{mb} MB zero_ (write): {t:.1f} us  {mb*1.048576e6/t/1e6:.2f} TB/s")
    t = bench(lambda: y.copy_(x)); print(f"{mb} MB copy (r+w): {t:.1f} us  {2*mb*1.048576e6/t/1e6:.2f} TB/s total")
    t = bench(lambda: ops.act_fwd(x, "silu")); print(f"{mb} MB fd_act silu (r+w): {t:.1f} us  {2*mb*1.048576e6/t/1e6:.2f} TB/s total")
    t = bench(lambda: x.sum()); print(f"{mb} MB sum (read): {t:.1f} us  {mb*1.048576e6/t/1e6:.2f} TB/s")
