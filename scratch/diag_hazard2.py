"""Race hunt, second pass: WHICH op of a timestep's backward first produces a different output under the three-stream schedule?
Every op wrapper of ops.py gets a device-side checksum of its outputs (int64 sum of the raw 16/32-bit words, written into a preallocated
buffer on the op's own stream); the per-(timestep, op index) checksums of the concurrent schedule are compared with those of the virtual
schedule (same buffers and order, one stream).   usage: FAIRDIFF_LIB=... python scratch/diag_hazard2.py [runs]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import finetune_fair_diffusion_amd  # noqa: F401,E402
import torch  # noqa: E402
import util_models as U  # noqa: E402
from finetune_fair_diffusion_amd import factory, layers, ops, unet as unet_mod  # noqa: E402
from finetune_fair_diffusion_amd.step import FairnessTrainer  # noqa: E402

dev = torch.device("cuda:0")
torch.set_num_threads(16)
sds = U.synthetic_sds(4, True, False, 80, 0.02, 0, "sd15")
pm = U.product_models(sds, dev, train_unet=True, train_te=False, size="sd15", eval_copies=True)
print(f"lib = {os.environ.get('FAIRDIFF_LIB', 'shipped')}", flush=True)
args = U.make_args(train_unet=True, train_text_encoder=False, size_face=224)
tr = FairnessTrainer(args, pm["text_encoder"], pm["unet"], pm["vae"], pm["classifier"], pm["scheduler"], eval_unet=pm["eval_unet"], device=dev)
tr.sync_and_update = lambda nb, apply=True: True
noises = torch.randn(8, 4, 64, 64, generator=torch.Generator().manual_seed(77))
tokens = factory.synthetic_tokens(77, 49408)

MAXOPS = 4096
CK = torch.zeros((20, MAXOPS), dtype=torch.int64, device=dev)
LOG = {}            # (timestep, index) -> description
STATE = dict(on=False, t=-1, n=0)


def rec(name, t):
    if not STATE["on"] or not torch.is_tensor(t) or not t.is_cuda or not t.is_contiguous() or t.numel() == 0:
        return
    i, n = STATE["t"], STATE["n"]
    if n >= MAXOPS:
        return
    STATE["n"] = n + 1
    w = t.view(torch.int32) if (t.element_size() * t.numel()) % 4 == 0 and t.element_size() in (2, 4) and (t.element_size() == 4 or t.shape[-1] % 2 == 0) else t.view(torch.int16) if t.element_size() == 2 else t
    CK[i, n:n + 1].copy_(w.sum(dtype=torch.int64).view(1))
    LOG[(i, n)] = f"{name} {tuple(t.shape)} {str(t.dtype).replace('torch.', '')}"


def wrap(mod, fname):
    fn = getattr(mod, fname)

    def w(*a, **k):
        out = fn(*a, **k)
        if isinstance(out, (tuple, list)):
            for j, o in enumerate(out):
                rec(f"{fname}[{j}]", o)
        else:
            rec(fname, out)
        return out
    setattr(mod, fname, w)


for f in ("gemm", "conv3x3", "conv_up2_bwd", "conv_small_cin", "groupnorm_bwd", "geglu_bwd_interleaved", "layernorm_bwd", "add", "attn_bwd", "downsum2x2"):
    wrap(ops, f)

orig_bs = tr.unet.backward_step
COUNTER = dict(i=0)


def backward_step(d_eps, gscale):
    STATE.update(on=True, t=COUNTER["i"], n=0)
    rec("d_eps", d_eps)
    try:
        return orig_bs(d_eps, gscale)
    finally:
        STATE["on"] = False
        COUNTER["i"] += 1


tr.unet.backward_step = backward_step


def run(virtual):
    tr.bwd_virtual = virtual
    COUNTER["i"] = 0
    CK.zero_()
    tr.train_step(tokens, noises, 20)
    torch.cuda.synchronize()
    return CK.cpu().clone()


run(False)
ref = run(True)
ref2 = run(True)
print("virtual schedule reproducible:", bool((ref == ref2).all()), "ops per timestep:", max(n for (_, n) in LOG) + 1, flush=True)
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    got = run(False)
    ne = (got != ref)
    print(f"--- concurrent run {r}: {int(ne.sum())} differing checksums")
    for i in range(20):
        idx = [j for j in ne[i].nonzero().view(-1).tolist() if not LOG[(i, j)].startswith(("attn_bwd[1]", "attn_bwd[2]"))]   # shared dK / dV accumulators: progress-dependent
        if not idx:
            continue
        nops = max(n for (t, n) in LOG if t == i) + 1
        f = idx[0]
        # sporadic (isolated) or propagating?
        print(f"  timestep {i:2d} (stream {i % 3}): first diff at op {f}/{nops}: {LOG[(i, f)]}; {len(idx)} of the following {nops - f} differ; next diffs {idx[1:8]}")
        for j in range(max(0, f - 6), min(nops, f + 4)):
            print(f"        op {j}: {LOG.get((i, j))}{'   <-- differs' if j in idx else ''}")
    sys.stdout.flush()
