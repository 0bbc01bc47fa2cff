"""How far ahead of the device can the host run on one HIP stream?  Times every C-ABI call of one frozen SD-v1.5 rollout after a sync:
calls return in ~4 us until the stream's queue is full, then each call waits for a slot."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finetune_fair_diffusion_amd import factory, ops

dev = torch.device("cuda:0")
args = factory.default_args(experiment="exp-1", train_unet=True, train_text_encoder=False, rank=4, train_images_per_prompt_GPU=8, train_GPU_batch_size=3,
                            val_GPU_batch_size=8, mixed_precision="fp16", weight_loss_img=0.0, weight_loss_face=0.0)
tr, models = factory.build_trainer(args, dev, cfgs=factory.SD15, seed=0, regularisers=False, lora_up_std=0.01)
tokens = factory.synthetic_tokens(13, factory.SD15["clip"].vocab_size)
enc = tr.encode_pair(tr.eval_te, tokens)
n8 = torch.randn(8, 4, 64, 64, device=dev)
tr.rollout(tr.eval_unet, enc, n8, 20); torch.cuda.synchronize()
lat = []
orig = ops._call
def timed(name, *a):
    t0 = time.perf_counter(); r = orig(name, *a); lat.append((time.perf_counter() - t0) * 1e6); return r
ops._call = timed
t0 = time.perf_counter()
tr.rollout(tr.eval_unet, enc, n8, 20)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
ops._call = orig
print(f"calls {len(lat)}  host {1e3 * (t1 - t0):.1f} ms  device {1e3 * (t2 - t0):.1f} ms")
import statistics
for lo in range(0, min(len(lat), 4000), 250):
    seg = lat[lo:lo + 250]
    print(f"calls {lo:5d}..{lo + len(seg):5d}: median {statistics.median(seg):6.1f} us  mean {statistics.mean(seg):6.1f} us  max {max(seg):7.1f} us  sum {sum(seg) / 1e3:6.2f} ms")
slow = [i for i, v in enumerate(lat) if v > 50]
print("first calls slower than 50 us:", slow[:20], " count:", len(slow), " their total:", round(sum(lat[i] for i in slow) / 1e3, 1), "ms of", round(sum(lat) / 1e3, 1))
