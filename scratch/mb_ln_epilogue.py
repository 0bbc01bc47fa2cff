"""Isolated cost of the LayerNorm epilogue: GEMM + standalone LayerNorm vs the GEMM that writes both (HIP events, hot caches)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops
dev = torch.device("cuda", 0)
def t(fn, n=40):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n
g = torch.Generator().manual_seed(0)
for (M, N, K) in [(65536, 320, 320), (32768, 320, 320)]:
    a = torch.randn(M, K, generator=g).to(dev).half(); b = (torch.randn(N, K, generator=g) * 0.1).to(dev).half(); r = torch.randn(M, N, generator=g).to(dev).half()
    bias = torch.randn(N, generator=g).to(dev); gm = torch.ones(N, device=dev); bt = torch.zeros(N, device=dev)
    for rep in range(2):
        ops.LN_EPILOGUE = False
        sep = t(lambda: ops.gemm(a, b, bias=bias, residual=r, ln=(gm, bt, 1e-5)))
        gem = t(lambda: ops.gemm(a, b, bias=bias, residual=r))
        ops.LN_EPILOGUE = True
        fus = t(lambda: ops.gemm(a, b, bias=bias, residual=r, ln=(gm, bt, 1e-5)))
        print(f"gemm {M}x{N}x{K} +res: GEMM alone {gem:.1f} us, GEMM + LayerNorm pass {sep:.1f} us, GEMM with the LayerNorm epilogue {fus:.1f} us")
