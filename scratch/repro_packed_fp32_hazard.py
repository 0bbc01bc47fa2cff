"""Stand-alone reproducer of the round-3 hazard (HISTORY.md "Rounds 2-4 at a glance"; DESIGN.md section 4 "build flags"): the four-rows-per-wave LayerNorm backward, built WITH hipcc's SLP
vectorisation (packed-fp32 VALU code), is run N times on fixed inputs on one stream while a second stream keeps the chip busy with the step's
GEMM / convolution / attention kernels; every output is compared bitwise with the result on an idle device.
    FAIRDIFF_LIB=<library built with -DFD_LN_BWD_MULTI_ROW and round-3 flags> python scratch/repro_packed_fp32_hazard.py [iterations]
With the shipped library (no packed-fp32 VALU) the count must be 0."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import finetune_fair_diffusion_amd  # noqa: F401,E402
import torch  # noqa: E402
from finetune_fair_diffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
F16 = ops.F16
g = torch.Generator().manual_seed(1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dev)


side = torch.cuda.Stream()
hog_x = rnd(16 * 64 * 64, 320).to(F16)
hog_w = rnd(320, 9 * 320, scale=0.02).to(F16)
hog_a = rnd(16384, 640).to(F16)
hog_b = rnd(640, 640, scale=0.04).to(F16)
hog_q = rnd(2 * 4096, 3 * 320).to(F16)


def hog(n):
    with torch.cuda.stream(side):
        for i in range(n):
            if i % 3 == 0:
                ops.conv3x3(hog_x, hog_w, 16, 64, 64)
            elif i % 3 == 1:
                ops.gemm(hog_a, hog_b)
            else:
                ops.attn_fwd(hog_q[:, :320], hog_q[:, 320:640], None, 2, 8, 4096, 4096, 40, 1, v=hog_q[:, 640:])


print("lib =", os.environ.get("FAIRDIFF_LIB", "shipped"), " iterations per shape:", N)
for (M, C) in ((16384, 640), (4096, 1280), (65536, 320)):
    x, dy, add = rnd(M, C).to(F16), rnd(M, C, scale=1e-3).to(F16), rnd(M, C, scale=1e-3).to(F16)
    gamma, beta = rnd(C).float() * 0.2 + 1.0, rnd(C).float() * 0.1
    _, st = ops.layernorm(x, gamma, beta, 1e-5, save_stats=True)
    torch.cuda.synchronize()
    ref = ops.layernorm_bwd(x, dy, gamma, st, add=add)
    torch.cuda.synchronize()
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    nel = torch.zeros((), dtype=torch.int64, device=dev)
    for it in range(N):
        if it % 8 == 0:
            hog(6)
        ne = ops.layernorm_bwd(x, dy, gamma, st, add=add) != ref
        bad += ne.any()
        nel += ne.sum()
    torch.cuda.synchronize()
    print(f"M={M} C={C}: {int(bad)} of {N} launches differ from the idle-device result ({int(nel)} elements)", flush=True)
