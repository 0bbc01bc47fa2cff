"""Would a row-concatenated frozen rollout (R1 and R2 through the same base-weight launches, M doubled) beat the two-stream schedule?
Times the frozen SD-v1.5 CFG rollout (20 DPM steps) at B = 8 and B = 16 on one stream, and two B = 8 rollouts on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finetune_fair_diffusion_amd import factory

dev = torch.device("cuda:0")
args = factory.default_args(experiment="exp-1", train_unet=True, train_text_encoder=False, rank=4, train_images_per_prompt_GPU=8, train_GPU_batch_size=3,
                            val_GPU_batch_size=16, mixed_precision="fp16", weight_loss_img=0.0, weight_loss_face=0.0)
tr, models = factory.build_trainer(args, dev, cfgs=factory.SD15, seed=0, regularisers=False, lora_up_std=0.01)
tokens = factory.synthetic_tokens(13, factory.SD15["clip"].vocab_size)
S = 20


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


enc = tr.encode_pair(tr.eval_te, tokens)
n8 = torch.randn(8, 4, 64, 64, device=dev)
n16 = torch.randn(16, 4, 64, 64, device=dev)


def one(noises):
    tr.rollout(tr.eval_unet, enc, noises, S)


def two():
    cur = torch.cuda.current_stream(); side = tr._side_stream()
    side.wait_stream(cur)
    ra, rb = {}, {}
    ga = tr.rollout_steps(tr.eval_unet, enc, n8, S, ra)
    with torch.cuda.stream(side):
        gb = tr.rollout_steps(tr.eval_unet, enc, n8, S, rb)
    for _ in ga:
        with torch.cuda.stream(side):
            next(gb, None)
    cur.wait_stream(side)


def many(k):
    """16 images as k rollouts of 16/k on k streams (stream 0 = current)"""
    cur = torch.cuda.current_stream()
    streams = [cur] + [tr._side_stream(i) for i in range(1, k)]
    parts = n16.chunk(k)
    gens = []
    for s, x in zip(streams, parts):
        if s is not cur:
            s.wait_stream(cur)
        with torch.cuda.stream(s):
            gens.append(tr.rollout_steps(tr.eval_unet, enc, x, S, {}))
    for _ in range(S + 1):
        for s, g in zip(streams, gens):
            with torch.cuda.stream(s):
                next(g, None)
    for s in streams[1:]:
        cur.wait_stream(s)


t8 = timed(lambda: one(n8)); t16 = timed(lambda: one(n16)); t2 = timed(two)
for k in (2, 4):
    print(f"16 images as {k} rollouts on {k} streams: {timed(lambda: many(k)):.1f} ms")
print(f"frozen rollout, S=20: B=8 one stream {t8:.1f} ms | B=16 one stream {t16:.1f} ms ({t16 / t8:.2f}x) | 2 x B=8 on two streams {t2:.1f} ms ({t2 / t8:.2f}x)")
