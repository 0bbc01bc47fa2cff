"""ViT-tail dense shapes (M ~ 2112): time per call and the tile the policy picks; run under FAIRDIFF_LIB=<bench lib> with FD_GEMM_T160=<n> to move the 128x160 threshold."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetune_fair_diffusion_amd import ops, lib
dev = torch.device("cuda:0")
def timeit(fn, n=50, rep=5):
    for _ in range(3): fn()
    ts = []
    for _ in range(rep):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1000 / n)
    return sorted(ts)[len(ts) // 2]
for M, N, K, res in [(2112, 1280, 1280, False), (2112, 1280, 1280, True), (2056, 1280, 1280, False), (2112, 768, 768, False), (2112, 768, 3072, True), (2112, 3072, 768, False), (2112, 5120, 1280, False), (2112, 1280, 5120, True),
                     (1024, 1280, 1280, False), (4096, 1280, 1280, False)]:
    a = (torch.randn(M, K, device=dev) * 0.5).half(); w = (torch.randn(N, K, device=dev) * K ** -0.5).half(); bias = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev).half() if res else None
    d = lib.GemmDesc(); d.M, d.N, d.K, d.batch, d.ldc, d.lda, d.ldb, d.alpha = M, N, K, 1, N, K, K, 1.0
    ws = ops.gemm_workspace(); d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    buf = ctypes.create_string_buffer(128); split = lib.get().fd_gemm_kernel_name(ctypes.byref(d), buf, 128)
    t = timeit(lambda: ops.gemm(a, w, bias=bias, residual=r))
    ref = (a.float() @ w.float().t() + bias + (r.float() if res else 0))
    err = float((ops.gemm(a, w, bias=bias, residual=r).float() - ref).abs().max() / ref.abs().max())
    print(f"{M} {N} {K} res={int(res)} {buf.value.decode()} split={split}: {t:.1f} us {2.0 * M * N * K / t / 1e6:.0f} TF err {err:.1e}", flush=True)
