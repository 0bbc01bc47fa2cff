"""soak of the configs[3] rollout length (S = 50, B = 8, exp-4: three attributes, OT targets) -- N consecutive training steps
through the product's train_step with the allocator's numbers logged per step: reserved / allocated / peak bytes, timesteps kept in HBM, the
number of allocator snapshot walks (``_usable_free_bytes`` refreshes when ``memory_reserved`` moves by > 1 GiB), OOM retries, ms per step.
usage: python scratch/soak_s50.py [steps] [collectives]
``collectives`` (round 5): initialise a single-rank RCCL process group and run every collective of the step through it
(FD_FORCE_COLLECTIVES), so the communicator's buffers are resident while the S = 50 chain fills the HBM; the driver-reported free memory and the
margin under 288 GiB are logged per step."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import finetune_fair_diffusion_amd  # noqa: F401,E402
import torch  # noqa: E402
from finetune_fair_diffusion_amd import factory  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
COLL = len(sys.argv) > 2 and sys.argv[2] == "collectives"
torch.cuda.set_device(0)
free0 = torch.cuda.mem_get_info()[0]
if COLL:
    import torch.distributed as dist
    os.environ["FD_FORCE_COLLECTIVES"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    t = torch.ones(1 << 20, device=dev)
    dist.all_reduce(t)
    dist.all_gather_into_tensor(torch.empty(1 << 20, device=dev), t)
    torch.cuda.synchronize()
    print(json.dumps(dict(rccl_resident_mib=round((free0 - torch.cuda.mem_get_info()[0] - (8 << 20)) / 2 ** 20, 1), note="driver-free memory taken by the communicator + its first collectives (two 4 MiB test tensors subtracted)")), flush=True)
S, B = 50, 8
args = factory.default_args(experiment="exp-4", train_unet=True, train_text_encoder=False, rank=4, train_images_per_prompt_GPU=B, train_GPU_batch_size=3,
                            val_GPU_batch_size=8, mixed_precision="fp16", size_face=224, img_size_small=224, weight_loss_img=8.0, weight_loss_face=1.0)
tr, _ = factory.build_trainer(args, dev, cfgs=factory.SD15, seed=0, regularisers=True, experiment="exp-4", lora_up_std=0.01)
tokens = factory.synthetic_tokens(13, 49408)
torch.manual_seed(5991)
rows = []
nxt = torch.randn(B, 4, 64, 64)
for step in range(N):
    noises, nxt = nxt, torch.randn(B, 4, 64, 64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = tr.train_step(tokens, noises, S, next_step=dict(tokens_ori=tokens, noises=nxt, S=S))
    torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    rows.append(dict(step=step, ms=round(1e3 * (time.perf_counter() - t0), 1), reserved_gib=round(torch.cuda.memory_reserved() / 2 ** 30, 2),
                     allocated_gib=round(torch.cuda.memory_allocated() / 2 ** 30, 2), peak_allocated_gib=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                     driver_free_gib=round(torch.cuda.mem_get_info()[0] / 2 ** 30, 2),
                     kept=min(S, 1 + max(tr.last_ctx_budget, 0)), snapshot_walks=getattr(tr, "_snap_walks", None), oom_retries=st.get("num_alloc_retries", 0),
                     finite=bool(out["grad_is_finite"])))
    print(json.dumps(rows[-1]), flush=True)
res = [r["reserved_gib"] for r in rows[5:]]
print(json.dumps(dict(summary=True, collectives=COLL, driver_free_min_gib=min(r["driver_free_gib"] for r in rows), steps=N, reserved_min=min(res), reserved_max=max(res), peak_allocated=max(r["peak_allocated_gib"] for r in rows),
                      oom_retries=rows[-1]["oom_retries"], ms_median=sorted(r["ms"] for r in rows[3:])[len(rows[3:]) // 2], kept=sorted(set(r["kept"] for r in rows)))))
