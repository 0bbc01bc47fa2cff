import sys; sys.path.insert(0, "/root/repo")
import torch
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda")
def bench(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (M, N) in [(65536, 320), (16384, 640), (4096, 1280), (1024, 1280), (26, 768)]:
    xs = [torch.randn(M, N, device=dev).half() for _ in range(6)]
    t = torch.randn(M, 8, device=dev).half()
    G = torch.zeros(N, 4, device=dev)
    i = [0]
    def f():
        i[0] += 1
        ops.lora_wgrad(xs[i[0] % 6], t, G, 4, 1, 4)
    us = bench(f)
    ref = (xs[0].float().t() @ t.float())[:, :4]
    G.zero_(); ops.lora_wgrad(xs[0], t, G, 4, 1, 4)
    err = float((G - ref).abs().max() / ref.abs().max())
    print(f"lora_wgrad M={M} N={N}: {us:.1f} us  ({M*N*2/us/1e6:.2f} TB/s)  rel err {err:.1e}")
