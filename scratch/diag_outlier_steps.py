"""Which phase holds the extra ~55 ms of the slow steps (5 of 20 in profiles/r04_bench_full_line_v2_20_steps_cfg1_cpu_protocol.json)?  N bench-style steps; the
HIP events of every step's phase marks are kept and evaluated after the last step (no extra sync inside the loop), next to the host time of each step,
the number of Python garbage collections and the allocator's counters."""
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import finetune_fair_diffusion_amd  # noqa: F401,E402
import torch  # noqa: E402
from finetune_fair_diffusion_amd import factory  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
if len(sys.argv) > 2 and sys.argv[2] == "nogc":
    gc.disable()
args = factory.default_args(experiment="exp-1", train_unet=True, train_text_encoder=False, rank=4, train_images_per_prompt_GPU=8, train_GPU_batch_size=3,
                            val_GPU_batch_size=8, mixed_precision="fp16", size_face=224, img_size_small=224, weight_loss_img=8.0, weight_loss_face=1.0)
tr, _ = factory.build_trainer(args, dev, cfgs=factory.SD15, seed=0, regularisers=True, lora_up_std=0.01)
tokens = factory.synthetic_tokens(13, 49408)
torch.manual_seed(5991)
nxt = torch.randn(8, 4, 64, 64)
marks, host, gcs, stats = [], [], [], []
for step in range(N + 3):
    noises, nxt = nxt, torch.randn(8, 4, 64, 64)
    tr.timers = True if step >= 3 else None
    g0 = sum(s["collections"] for s in gc.get_stats())
    t0 = time.perf_counter()
    tr.train_step(tokens, noises, 20, next_step=dict(tokens_ori=tokens, noises=nxt, S=20))
    if step >= 3:
        host.append(1e3 * (time.perf_counter() - t0))
        marks.append(list(tr._marks))
        gcs.append(sum(s["collections"] for s in gc.get_stats()) - g0)
        st = torch.cuda.memory_stats()
        stats.append((st.get("num_alloc_retries", 0), st.get("num_device_alloc", 0), round(torch.cuda.memory_reserved() / 2 ** 30, 2), getattr(tr, "_snap_walks", 0)))
torch.cuda.synchronize()
med = sorted(host)[len(host) // 2]
for i, (h, m) in enumerate(zip(host, marks)):
    ph = {}
    for (n0, e0), (_, e1) in zip(m[:-1], m[1:]):
        ph[n0] = round(ph.get(n0, 0.0) + e0.elapsed_time(e1), 1)
    flag = "  <-- slow" if h > med + 25 else ""
    print(f"step {i:2d} host {h:7.1f} ms  gc {gcs[i]}  alloc(retries, device_allocs, reserved GiB, snapshot walks) {stats[i]}  {json.dumps(ph)}{flag}")
