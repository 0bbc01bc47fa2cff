import sys, time; sys.path.insert(0, "/root/repo")
import torch
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda")
a = torch.randn(1024, 64, device=dev).half(); b = torch.randn(64, 64, device=dev).half()
g = torch.ones(64, device=dev); be = torch.zeros(64, device=dev)
for name, f in [("add", lambda: ops.add(a, a)), ("gemm", lambda: ops.gemm(a, b)), ("layernorm", lambda: ops.layernorm(a, g, be)), ("act", lambda: ops.act_fwd(a, "silu"))]:
    for _ in range(100): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5000
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: host {1e6*(t1-t0)/n:.1f} us/call, incl. drain {1e6*(t2-t0)/n:.1f} us/call")
