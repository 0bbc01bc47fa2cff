"""Headline benchmark: trained images / second of the fairness-finetuning step (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full training step of exp-1-debias-gender/1-main-debias.py:1746-2029 (minus wandb/plots
and the detector side-car): R1 + R2 no-grad CFG rollouts of B images, R3 rollout with
recompute-backward, classifier, VAE forward/backward, LoRA-gradient all-reduce, AdamW + EMA.
Workload = BASELINE.json configs[1]: exp-1-debias-gender, batch 8 per GPU, S=20 DPM-Solver++ steps,
LoRA rank 4 on the U-Net, fp16, SD-v1.5 512x512, synthetic weights / prompts / noise.
One process per GPU; weak scaling (each rank trains its own B images, one RCCL all-reduce of the flat
LoRA-gradient buffer per step).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_UNET = 0.798e12      # algorithmic FLOP per U-Net sample-forward (SURVEY.md 8d / BASELINE.md 2)
F_VAE = 2.515e12       # per 512x512 image decode
MFMA_PEAK_F16 = 2.5e15  # dense fp16 MFMA peak, MI355X_MICROARCH.md


def f_img(S):
    return 8 * S * F_UNET + 4 * F_VAE


def pmc_traffic(kernel, algorithmic_bytes_per_launch):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE / WRITE_SIZE cannot be read from
    inside a process; they are collected with rocprofv3 --pmc in separate runs on the kernel's own shapes, profiles/r01_pmc_traffic.json,
    with the gfx950 corrections of MI355X_MICROARCH.md applied).  This run's per-launch figure = its mean ALGORITHMIC bytes per launch
    (exact, from every launch's M, N, K) x the measured traffic / algorithmic ratio of that kernel variant."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None, "no PMC summary committed"
    d = json.load(open(path))["kernels"].get(kernel)
    if d is None:
        return None, "no PMC summary for " + kernel
    ratio = sum(x["hbm_bytes"] for x in d["shapes"]) / sum(x["algorithmic_bytes"] for x in d["shapes"])
    return algorithmic_bytes_per_launch * ratio, "bytes/launch = mean algorithmic bytes/launch of this run (%.1f MB) x %.2f, the rocprofv3 PMC ratio " \
        "(FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) / algorithmic bytes on this kernel's shapes; Infinity-Cache hits are included" % (
            algorithmic_bytes_per_launch / 1e6, ratio)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="train_images_per_prompt_GPU")
    ap.add_argument("--S", type=int, default=20, help="denoising steps")
    ap.add_argument("--rank", type=int, default=4, help="LoRA rank")
    ap.add_argument("--tiny", action="store_true", help="tiny model config (plumbing check only; not a valid bench line)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU plumbing checks)")
    ap.add_argument("--share_gpu0", action="store_true", help="plumbing check: every rank uses cuda:0")
    ap.add_argument("--no_regularisers", action="store_true", help="drop the CLIP/DINOv2 image-semantics and SFNet face-realism terms (loss_fair only)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_roofline", action="store_true")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    if a.share_gpu0:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    from finetune_fair_diffusion_amd import factory, ops
    args = factory.default_args(train_unet=True, train_text_encoder=False, rank=a.rank, train_images_per_prompt_GPU=a.batch,
                                train_GPU_batch_size=3, val_GPU_batch_size=8, mixed_precision="fp16",
                                size_face=64 if a.tiny else 224, img_size_small=56 if a.tiny else 224,
                                weight_loss_img=0.0 if a.no_regularisers else 8.0,   # debias-unet.yaml:4
                                weight_loss_face=0.0 if a.no_regularisers else 1.0)  # debias-unet.yaml:5
    cfgs = factory.TINY if a.tiny else factory.SD15
    tr, models = factory.build_trainer(args, dev, cfgs=cfgs, seed=0, rank=rank, world_size=world, regularisers=not a.no_regularisers,
                                       lora_up_std=0.01)   # up != 0 as after one warm-up optimiser step
    L = 13
    tokens = factory.synthetic_tokens(L, cfgs["clip"].vocab_size)
    hw = cfgs["unet"].sample_size
    torch.manual_seed(5991 + rank)  # set_seed(seed, device_specific=True) (:693): per-rank noise, drawn on the CPU (:1746-1749)

    def one_step():
        noises = torch.randn([a.batch, 4, hw, hw], dtype=torch.float32)
        return tr.train_step(tokens, noises.to(dev), a.S)

    def fence():
        torch.cuda.synchronize()      # this rank's work is done before it enters the barrier ...
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()  # ... and the barrier's own collective has completed before the clock is read

    for _ in range(a.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = one_step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = 1e3 * dt / a.steps
    value = world * a.batch * a.steps / dt

    line = {
        "metric": "training-images/sec (SD-v1.5 512^2, 20-step DPM-Solver++ unroll, fairness-finetune step)",
        "value": value, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": "exp-1-debias-gender, batch %d/GPU, %d denoising steps, LoRA rank %d on U-Net, fp16, SD-v1.5 512x512 (BASELINE configs[1])"
                               % (a.batch, a.S, a.rank) if not a.tiny else "TINY plumbing config (not a bench line)",
                   "global_batch": world * a.batch, "steps_per_s": a.steps / dt, "parallelism": f"dp{world}",
                   "algorithmic_flop_per_image": f_img(a.S), "step_mfma_frac": value / world * f_img(a.S) / MFMA_PEAK_F16,
                   "loss_fair_mean": float(out["loss_fair"][out["loss_fair"] != -1].mean()) if (out["loss_fair"] != -1).any() else None,
                   "loss_terms": "loss_fair" if a.no_regularisers else "loss_fair + 8*dyn*(loss_CLIP[ViT-H/14] + loss_DINO[ViT-B/14]) + 1*loss_face[SFNet-20]",
                   "loss_CLIP_mean": float(out["loss_CLIP"].mean()) if "loss_CLIP" in out else None,
                   "loss_DINO_mean": float(out["loss_DINO"].mean()) if "loss_DINO" in out else None,
                   "loss_face_mean": float(out["loss_face"][out["loss_face"] != -1].mean()) if "loss_face" in out and (out["loss_face"] != -1).any() else None,
                   "grad_is_finite": bool(out["grad_is_finite"]),
                   "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                   "r3_activation_gb_per_timestep": round(tr.last_ctx_bytes / 2 ** 30, 2),
                   "r3_timesteps_kept_in_hbm": min(a.S, 1 + max(tr.last_ctx_budget, 0)) if tr.keep_activations else 0},
    }

    if rank == 0 and not a.no_roofline:
        # roofline pass: one more identical step with per-launch HIP events on the GEMM/conv kernel family
        ops.TIMER = ops.OpTimer()
        one_step()
        summ = ops.TIMER.summary()
        ops.TIMER = None
        top = max(summ.items(), key=lambda kv: kv[1]["ms"])
        name, s = top
        achieved = s["flops"] / (s["ms"] * 1e-3) / 1e12
        traffic, traffic_note = pmc_traffic(name, s["bytes"] / s["launches"])
        line["roofline"] = {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": MFMA_PEAK_F16 / 1e12, "unit": "TFLOP/s",
                            "frac": achieved / (MFMA_PEAK_F16 / 1e12), "traffic": traffic, "traffic_note": traffic_note,
                            "launches": s["launches"],
                            "avg_launch_us": 1e3 * s["ms"] / s["launches"], "algorithmic_flop_per_launch": s["flops"] / s["launches"],
                            "algorithmic_bytes_per_launch": s["bytes"] / s["launches"],
                            # the same launches against the other roof: short-K shapes (K = 320) of this family are closer to HBM than to MFMA
                            "algorithmic_GBps": s["bytes"] / (s["ms"] * 1e-3) / 1e9, "hbm_frac_of_8TBps": s["bytes"] / (s["ms"] * 1e-3) / 8e12,
                            "family": {k: {"launches": v["launches"], "ms": round(v["ms"], 2), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)}
                                       for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}}
    if world == 1 and not a.no_roofline:
        # informational only (NOT the headline value): the same step when R3 consumes R1's recorded forward instead of
        # recomputing the bit-identical rollout; measured after the timed region
        tr.share_r1_r3 = True
        one_step()
        fence()
        t1 = time.perf_counter()
        one_step()
        fence()
        line["config"]["images_per_s_if_r3_shares_r1_forward"] = a.batch / (time.perf_counter() - t1)
        tr.share_r1_r3 = False
    if world > 1:
        dist.barrier()

    if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.tiny:
        line["cpu_baseline"] = cpu_baseline(a.S)

    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(S):
    """The oracle (fp32 PyTorch restatement of the reference's diffusers path) timed on this box's host
    cores on a bounded sample: one CFG pair (batch 2) of the SD-v1.5 U-Net forward -- the op that is
    >95 % of the step's FLOPs -- extrapolated to images/s with the algorithmic FLOPs per image."""
    import torch
    from oracle import nn_unet
    cores = min(os.cpu_count() or 1, 32)  # beyond ~32 threads the fp32 convs of a batch-2 call stop scaling
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    with torch.no_grad():
        unet = nn_unet.UNet2DConditionModel(nn_unet.UNetConfig())
        x = torch.randn(2, 4, 64, 64)
        enc = torch.randn(2, 13, 768)
        t0 = time.perf_counter()
        n = 0
        while n < 1 or (time.perf_counter() - t0 < 15.0 and n < 8):
            unet(x, torch.tensor(500), enc)
            n += 1
        dt = (time.perf_counter() - t0) / n
    flops_per_s = 2 * F_UNET / dt
    return {"value": flops_per_s / f_img(S), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 U-Net forward, SD-v1.5 size, CFG pair (batch 2), {n} calls in {dt * n:.1f} s: {dt:.2f} s per call = {flops_per_s / 1e12:.3f} TFLOP/s on {cores} threads; "
                      f"extrapolated with {f_img(S) / 1e12:.1f} algorithmic TFLOP per trained image"}


if __name__ == "__main__":
    main()
