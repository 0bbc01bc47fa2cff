"""PMC targets (round 5: one shape per (kernel, grid) so that per-kernel counter means are per-shape): the GEMM / conv kernels that lead the bench's kernel-time ranking, on their U-Net shapes at CFG batch 16.
Buffers are rotated over enough sets to exceed the 256 MiB Infinity Cache so that FETCH_SIZE reflects memory-side traffic.  Writes a
manifest (launch order, kernel name from fd_gemm_kernel_name, algorithmic bytes) that scratch/pmc_r03_summary.py joins with the counters."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from finetune_fair_diffusion_amd import lib, ops
dev = torch.device("cuda")
manifest = []

def kname(d):
    buf = ctypes.create_string_buffer(128)
    split = lib.get().fd_gemm_kernel_name(ctypes.byref(d), buf, 128)
    return buf.value.decode(), max(split, 1)

for (B, H, C) in ((16, 64, 320), (16, 32, 640), (16, 16, 1280), (16, 8, 1280)):      # square convs; the ping-pong kernels take the first three
    nset = max(8, int(300e6 / (B * H * H * C * 2 * 2)) + 1)
    xs = [torch.randn(B * H * H, C, device=dev).half() for _ in range(nset)]
    w = (torch.randn(C, 9 * C, device=dev) * 0.02).half()
    bias = torch.randn(C, device=dev)
    outs = [torch.empty(B * H * H, C, device=dev, dtype=torch.float16) for _ in range(nset)]
    d = lib.GemmDesc(); d.M, d.N, d.K, d.batch, d.conv, d.conv_mode, d.Bn, d.H, d.W, d.Cin, d.Ho, d.Wo = B * H * H, C, 9 * C, 1, 1, 0, B, H, H, C, H, H
    ws = ops.gemm_workspace(); d.workspace, d.workspace_bytes, d.ldc = ws.data_ptr(), ws.numel() * 4, C
    name, split = kname(d)
    for i in range(2 * nset):
        ops.conv3x3(xs[i % nset], w, B, H, H, bias=bias, out=outs[i % nset])
    M, N, K = B * H * H, C, 9 * C
    manifest.append(dict(kernel=name, split=split, kind="conv3x3", B=B, H=H, C=C, M=M, N=N, K=K, launches=2 * nset,
                         algorithmic_bytes=2.0 * (M * C + N * K + M * N)))
    torch.cuda.synchronize()
    del xs, outs
for (M, N, K) in ((65536, 2560, 320), (65536, 320, 320), (65536, 960, 320), (16384, 640, 640), (16384, 1920, 640), (4096, 1280, 1280), (4096, 3840, 1280), (4096, 1280, 5120)):
    nset = max(8, int(300e6 / ((M * K + M * N) * 2)) + 1)
    As = [torch.randn(M, K, device=dev).half() for _ in range(nset)]
    b = (torch.randn(N, K, device=dev) * 0.05).half()
    outs = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(nset)]
    d = lib.GemmDesc(); d.M, d.N, d.K, d.batch = M, N, K, 1
    ws = ops.gemm_workspace(); d.workspace, d.workspace_bytes, d.ldc = ws.data_ptr(), ws.numel() * 4, N
    name, split = kname(d)
    for i in range(2 * nset):
        ops.gemm(As[i % nset], b, out=outs[i % nset])
    manifest.append(dict(kernel=name, split=split, kind="dense", M=M, N=N, K=K, launches=2 * nset, algorithmic_bytes=2.0 * (M * K + N * K + M * N)))
    torch.cuda.synchronize()
    del As, outs
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(manifest, open(os.path.join(ROOT, "gpurun_out", "pmc_r05_manifest.json"), "w"), indent=1)
print("done", len(manifest))
