"""Race hunt for the three-stream backward.  One B = 8 / S = 20 SD-v1.5-size step per variant; the per-stream gradient
buffers of every variant are compared BITWISE with the ``virtual`` schedule (same buffer partition, all on one stream: same fp32 order).
usage: FAIRDIFF_LIB=... python scratch/diag_hazard.py variant [variant ...]
variants: s3 (shipped), s2, v3 (virtual), keep (s3 + contexts kept alive), sync (s3 + device sync after every timestep's enqueue),
          delay (s3 + random stream delays), s3x<N> (N repeats)"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import finetune_fair_diffusion_amd  # noqa: F401,E402
import torch  # noqa: E402
import util_models as U  # noqa: E402
from finetune_fair_diffusion_amd import ops  # noqa: E402
from finetune_fair_diffusion_amd.step import FairnessTrainer  # noqa: E402

dev = torch.device("cuda:0")
t0 = time.time()
torch.set_num_threads(16)
sds = U.synthetic_sds(4, True, False, 80, 0.02, 0, "sd15")
pm = U.product_models(sds, dev, train_unet=True, train_te=False, size="sd15", eval_copies=True)
print(f"models built in {time.time() - t0:.1f} s; lib = {os.environ.get('FAIRDIFF_LIB', 'shipped')}", flush=True)
args = U.make_args(train_unet=True, train_text_encoder=False, size_face=224)
tr = FairnessTrainer(args, pm["text_encoder"], pm["unet"], pm["vae"], pm["classifier"], pm["scheduler"], eval_unet=pm["eval_unet"], device=dev)
tr.sync_and_update = lambda nb, apply=True: True
bank = tr.banks[0]
noises = torch.randn(8, 4, 64, 64, generator=torch.Generator().manual_seed(77))
from finetune_fair_diffusion_amd import factory  # noqa: E402
tokens = factory.synthetic_tokens(77, 49408)


def run(nst=3, virtual=False, keep=False, sync=False, delay=None):
    tr.bwd_streams, tr.bwd_virtual, tr.bwd_keep_alive = nst, virtual, keep
    tr.debug_partials = []
    ops.DELAY = delay
    orig = tr.unet.backward_step
    if sync:
        def bs(*a, **k):
            orig(*a, **k)
            torch.cuda.synchronize()
        tr.unet.backward_step = bs
    try:
        out = tr.train_step(tokens, noises, 20)
        torch.cuda.synchronize()
    finally:
        tr.unet.backward_step = orig
        ops.DELAY = None
    return out, tr.debug_partials[0]


def family(n):
    return "kv" if (".attn2.processor.to_k_lora." in n or ".attn2.processor.to_v_lora." in n) else "rest"


def compare(name, parts, ref):
    for k, (a, b) in enumerate(zip(parts, ref)):
        bad = []
        for n in bank.names:
            x, y = bank.view(n, a), bank.view(n, b)
            if not torch.equal(x, y):
                bad.append((n, float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))))
        rest = [(n, e) for n, e in bad if family(n) == "rest"]
        kv = [(n, e) for n, e in bad if family(n) == "kv"]
        print(f"[{name}] buffer {k}: {len(rest)} 'rest' tensors differ (max rel {max([e for _, e in rest], default=0):.2e}), "
              f"{len(kv)} kv tensors differ (max rel {max([e for _, e in kv], default=0):.2e})")
        for n, e in rest[:12]:
            print(f"      {n}  {e:.2e}")
        if len(rest) > 12:
            print(f"      ... first in bank order {rest[0][0]}, last {rest[-1][0]}")
    sys.stdout.flush()


run()   # warm-up (allocator growth, lazily built operands)
_, ref3 = run(3, virtual=True)
_, ref3b = run(3, virtual=True)
compare("virtual-3 twice (must be 0 apart from kv... and kv too: one stream)", ref3b, ref3)
refs = {3: ref3}
for v in sys.argv[1:]:
    rep = 1
    if "x" in v:
        v, r = v.split("x")
        rep = int(r)
    for j in range(rep):
        if v == "s3":
            _, p = run(3)
        elif v == "s2":
            if 2 not in refs:
                _, refs[2] = run(2, virtual=True)
            _, p = run(2)
            compare(f"s2 #{j}", p, refs[2])
            continue
        elif v == "keep":
            _, p = run(3, keep=True)
        elif v == "sync":
            _, p = run(3, sync=True)
        elif v == "delay":
            _, p = run(3, delay=(0.02, 200000, random.Random(j)))
        elif v == "v3":
            _, p = run(3, virtual=True)
        else:
            raise SystemExit(f"unknown variant {v}")
        compare(f"{v} #{j}", p, ref3)
