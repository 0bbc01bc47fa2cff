"""Per-step host time next to caching-allocator activity (new segments = hipMalloc calls, frees) for 8 training steps of the bench workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finetune_fair_diffusion_amd import factory
dev = torch.device("cuda:0")
args = factory.default_args(experiment="exp-1", train_unet=True, train_text_encoder=False, rank=4, train_images_per_prompt_GPU=8, train_GPU_batch_size=3,
                            val_GPU_batch_size=8, mixed_precision="fp16", weight_loss_img=8.0, weight_loss_face=1.0)
tr, models = factory.build_trainer(args, dev, cfgs=factory.SD15, seed=0, regularisers=True, lora_up_std=0.01)
tokens = factory.synthetic_tokens(13, factory.SD15["clip"].vocab_size)
torch.manual_seed(5991)
prev = None
for i in range(9):
    noises = torch.randn([8, 4, 64, 64])
    t0 = time.perf_counter()
    tr.train_step(tokens, noises.to(dev), 20)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    s = torch.cuda.memory_stats()
    cur = (s["segment.all.allocated"], s["segment.all.freed"], s["num_alloc_retries"], s["allocation.all.allocated"])
    d = tuple(c - p for c, p in zip(cur, prev)) if prev else cur
    prev = cur
    print(f"step {i}: {dt:7.1f} ms  reserved {s['reserved_bytes.all.current'] / 2**30:6.1f} GiB  peak alloc {s['allocated_bytes.all.peak'] / 2**30:6.1f} GiB  "
          f"new segments {d[0]:4d}  freed segments {d[1]:4d}  retries {d[2]}  tensor allocations {d[3]}")
