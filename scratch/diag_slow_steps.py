"""Which steps are slow (+80 ms in ~1 of 5 on the pool's boxes) and what coincides with them?  30 steps of the bench configuration, every step fenced; per step: GPU phase
times, host enqueue time per phase, caching-allocator counters, Python GC activity, and the device's clocks / power read from sysfs right after the step."""
import gc, glob, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finetune_fair_diffusion_amd  # noqa
import torch
from finetune_fair_diffusion_amd import factory

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
args = factory.default_args(experiment="exp-1", train_unet=True, train_text_encoder=False, rank=4, train_images_per_prompt_GPU=8, train_GPU_batch_size=3,
                            val_GPU_batch_size=8, mixed_precision="fp16", size_face=224, img_size_small=224, weight_loss_img=8.0, weight_loss_face=1.0)
cfgs = factory.SD15
tr, models = factory.build_trainer(args, dev, cfgs=cfgs, seed=0, rank=0, world_size=1, regularisers=True, experiment="exp-1", lora_up_std=0.01)
tokens = factory.synthetic_tokens(13, cfgs["clip"].vocab_size)
hw = cfgs["unet"].sample_size
torch.manual_seed(5991)
nxt = [torch.randn([8, 4, hw, hw])]
gcs = []
gc.callbacks.append(lambda phase, info: gcs.append((phase, info.get("generation"), time.perf_counter())) if phase == "stop" else None)


def sysfs():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"):
            try:
                cur = [l for l in open(os.path.join(card, name)).read().splitlines() if l.rstrip().endswith("*")]
                if cur:
                    out[name] = cur[0].strip()
            except OSError:
                pass
        for hw_ in glob.glob(os.path.join(card, "hwmon/hwmon*")):
            for name in ("power1_average", "power1_input", "temp1_input", "temp2_input", "freq1_input"):
                try:
                    out[name] = int(open(os.path.join(hw_, name)).read())
                except (OSError, ValueError):
                    pass
        if out:
            break
    return out


def one_step():
    noises = nxt[0]
    nxt[0] = torch.randn([8, 4, hw, hw])
    return tr.train_step(tokens, noises, 20, next_step=dict(tokens_ori=tokens, noises=nxt[0], S=20))


import threading
samples, stop = [], [False]


def sampler():
    paths = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        hm = glob.glob(os.path.join(card, "hwmon/hwmon*"))
        if hm and os.path.exists(os.path.join(hm[0], "freq1_input")):
            paths = dict(f=os.path.join(hm[0], "freq1_input"), p=os.path.join(hm[0], "power1_average"), t=os.path.join(hm[0], "temp1_input"))
            break
    while not stop[0] and paths:
        try:
            samples.append((time.perf_counter(), int(open(paths["f"]).read()) // 1000000, int(open(paths["p"]).read()) // 1000000))
        except (OSError, ValueError):
            pass
        time.sleep(0.02)


FIXED = "fixed" in sys.argv
if FIXED:
    fixed_noise = torch.randn([8, 4, hw, hw])

    def one_step():      # noqa: F811
        return tr.train_step(tokens, fixed_noise, 20, next_step=dict(tokens_ori=tokens, noises=fixed_noise, S=20))
for _ in range(3):
    one_step()
threading.Thread(target=sampler, daemon=True).start()
tr.timers = True
torch.cuda.synchronize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
nofence = 'nofence' in sys.argv
prev = torch.cuda.memory_stats()
rows = []
for i in range(N):
    g0 = len(gcs)
    t0 = time.perf_counter()
    one_step()
    t1 = time.perf_counter()
    if not nofence:
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    st = torch.cuda.memory_stats()
    sm = [x for x in samples if t0 <= x[0] <= t0 + 0.7]         # the rollout phase
    row = dict(step=i, sclk_rollout=(min(x[1] for x in sm), sum(x[1] for x in sm) // max(len(sm), 1), max(x[1] for x in sm)) if sm else None,
               watts_rollout=(min(x[2] for x in sm), sum(x[2] for x in sm) // max(len(sm), 1), max(x[2] for x in sm)) if sm else None,
               ms=round((t2 - t0) * 1e3, 1), enqueue_ms=round((t1 - t0) * 1e3, 1), phases=tr.phase_ms() if not nofence else None, host=tr.host_phase_ms(),
               alloc_retries=st["num_alloc_retries"] - prev["num_alloc_retries"], device_allocs=st.get("num_device_alloc", 0) - prev.get("num_device_alloc", 0),
               device_frees=st.get("num_device_free", 0) - prev.get("num_device_free", 0), reserved_gb=round(st["reserved_bytes.all.current"] / 2 ** 30, 2),
               gc=[(g[1]) for g in gcs[g0:]], snap_walks=getattr(tr, "_snap_walks", None), sysfs=sysfs())
    prev = st
    rows.append(row)
    print(json.dumps(row), flush=True)
ms = sorted(r["ms"] for r in rows)
med = ms[len(ms) // 2]
print("median", med, "slow steps (> median + 40 ms):", [r["step"] for r in rows if r["ms"] > med + 40])
