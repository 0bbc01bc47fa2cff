"""hipGraph replay of the frozen model's forward (FD_R2_GRAPH=1, unet.GraphedForward): images of the frozen side and the whole LoRA gradient must be
BIT-identical to the eager schedule; prints both and the step times."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import finetune_fair_diffusion_amd  # noqa: F401,E402
import torch  # noqa: E402
from finetune_fair_diffusion_amd import factory  # noqa: E402

dev = torch.device("cuda:0")
args = factory.default_args(experiment="exp-1", train_unet=True, train_text_encoder=False, rank=4, train_images_per_prompt_GPU=8, train_GPU_batch_size=3,
                            val_GPU_batch_size=8, mixed_precision="fp16", size_face=224, img_size_small=224, weight_loss_img=8.0, weight_loss_face=1.0)
tr, _ = factory.build_trainer(args, dev, cfgs=factory.SD15, seed=0, regularisers=True, lora_up_std=0.01)
grads = {}
tr.sync_and_update = lambda nb, apply=True: (grads.__setitem__(0, tr.banks[0].grad.clone()), True)[1]
tokens = factory.synthetic_tokens(13, 49408)
g = torch.Generator().manual_seed(3)
noise = [torch.randn(8, 4, 64, 64, generator=g) for _ in range(4)]
res = {}
for mode in (False, True, False, True):
    tr.r2_graph = mode
    tr._r2_pre = None
    outs = []
    for i in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = tr.train_step(tokens, noise[i], 20, next_step=dict(tokens_ori=tokens, noises=noise[i + 1], S=20))
        torch.cuda.synchronize()
        outs.append((out["images_ori"].clone(), grads[0].clone(), 1e3 * (time.perf_counter() - t0), tr.last_r2_prefetched))
    print(f"r2_graph={mode}: step ms {[round(o[2], 1) for o in outs]}  prefetched {[o[3] for o in outs]}", flush=True)
    res.setdefault(mode, outs)
for i in range(3):
    a, b = res[False][i], res[True][i]
    print(f"step {i}: images_ori equal {torch.equal(a[0], b[0])}  gradient equal {torch.equal(a[1], b[1])}")
