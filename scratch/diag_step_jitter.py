"""Why does every other step take ~50 ms longer?  Per-step phase times (HIP events), host phase times and caching-allocator counters for 10
consecutive bench steps."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from finetune_fair_diffusion_amd import factory
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
args = factory.default_args(experiment="exp-1", train_unet=True, train_text_encoder=False, rank=4, train_images_per_prompt_GPU=8, train_GPU_batch_size=3,
                            val_GPU_batch_size=8, mixed_precision="fp16", size_face=224, img_size_small=224, weight_loss_img=8.0, weight_loss_face=1.0)
tr, models = factory.build_trainer(args, dev, cfgs=factory.SD15, seed=0, rank=0, world_size=1, regularisers=True, experiment="exp-1", lora_up_std=0.01)
tokens = factory.synthetic_tokens(13, 49408)
torch.manual_seed(5991)
tr.timers = True
keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams")
import gc
if os.environ.get("DIAG_NOGC"):
    gc.disable()
prev = None
for i in range(12):
    noises = torch.randn(8, 4, 64, 64)
    t0 = time.perf_counter()
    tr.train_step(tokens, noises, 20)
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0)
    st = torch.cuda.memory_stats()
    cur = {k: st.get(k, 0) for k in keys}
    ph = {k: round(v, 1) for k, v in tr.phase_ms().items()}
    hp = {k: round(v, 1) for k, v in tr.host_phase_ms().items()}
    print(f"step {i}: {dt:7.1f} ms  gc={gc.get_count()}  alloc-delta={ {k: cur[k] - (prev or cur)[k] for k in keys} }  reserved={st['reserved_bytes.all.current'] / 2**30:.1f} GiB")
    print("   dev ", ph)
    print("   host", hp, flush=True)
    prev = cur
