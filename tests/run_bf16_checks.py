"""bf16 working-dtype checks (BASELINE configs[4]) -- run as a SCRIPT in its own process because the working dtype of a process is fixed
at import (FD_DTYPE=bf16 selects libfairdiff_hip_bf16.so); tests/test_bf16_gpu.py launches it.

Bands (bf16 has 8 significand bits against fp16's 11, i.e. 8x the rounding step; the reference itself never ran bf16):
  kernels vs fp32 torch           2e-2 of max|ref|   (fp16 library: 2e-3 .. 5e-3)
  tiny U-Net eps vs fp32 oracle   8e-2               (fp16: 2e-2);  LoRA gradients per family 2e-1 (fp16: 5e-2), cosine > 0.995
  SD-v1.5-size U-Net eps          4e-2 (relative RMS 2e-2)
  full tiny training step         images 1e-1, exact targets, loss_fair 5e-2, end-to-end gradient cosine > 0.8 (measured 0.87; ReLU / clamp mask flips)
  exp-3 step on the d=40 model    (gender x race: 6-logit head, both sides on the oracle's OT targets)  images 1.5e-1 / RMS 4e-2, per-attribute
                                  loss_fair 8e-2, end-to-end gradient cosine > 0.80 (measured 0.87 in rounds 3-5; the gate leaves 0.07 for rounding-only changes of the forward)
"""
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
assert os.environ.get("FD_DTYPE") == "bf16", "run with FD_DTYPE=bf16"
import util_models as U  # noqa: E402
from finetune_fair_diffusion_amd import lib, ops  # noqa: E402

dev = torch.device("cuda:0")
assert lib.get().fd_working_dtype().decode() == "bf16" and ops.F16 == torch.bfloat16
BF = torch.bfloat16


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-20))


def check(name, a, b, tol):
    e = relerr(a, b)
    print(f"[bf16: {name}] rel max err {e:.3e} (tol {tol:.1e})")
    assert math.isfinite(e) and e <= tol, f"{name}: {e} > {tol}"


def rnd(*shape, seed, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(dev).to(BF)


def kernels():
    a, b = rnd(4096, 1280, seed=1), rnd(320, 1280, seed=2)
    check("gemm 4096x320x1280", ops.gemm(a, b), a.float() @ b.float().t(), 1e-2)
    a, b = rnd(65536, 320, seed=3), rnd(320, 320, seed=4)
    bias = torch.randn(320, generator=torch.Generator().manual_seed(5)).to(dev)
    res = rnd(65536, 320, seed=6)
    check("gemm 65536x320x320 + bias + residual (256x320 tile)", ops.gemm(a, b, bias=bias, residual=res), a.float() @ b.float().t() + bias + res.float(), 1e-2)
    x = rnd(2 * 32 * 32, 320, seed=7)
    w = rnd(320, 9 * 320, seed=8, scale=0.02)
    ref = F.conv2d(x.float().reshape(2, 32, 32, 320).permute(0, 3, 1, 2), w.float().reshape(320, 3, 3, 320).permute(0, 3, 1, 2), padding=1)
    y, _, _ = ops.conv3x3(x, w, 2, 32, 32)
    check("conv3x3 320->320 @32^2", y.reshape(2, 32, 32, 320).permute(0, 3, 1, 2), ref, 1e-2)
    g, st = ops.groupnorm(x, None, 2, 1024, 32, 1e-5, torch.ones(320, device=dev), torch.zeros(320, device=dev), True)
    check("groupnorm+silu", g.reshape(2, 1024, 320), F.silu(F.group_norm(x.float().reshape(2, 1024, 320).permute(0, 2, 1), 32, eps=1e-5)).permute(0, 2, 1), 2e-2)
    n = ops.layernorm(x, torch.ones(320, device=dev), torch.zeros(320, device=dev))
    check("layernorm", n, F.layer_norm(x.float(), (320,)), 2e-2)
    B, H, T, d = 2, 8, 1024, 40
    C = H * d
    q, k, v = rnd(B, T, C, seed=11), rnd(B, T, C, seed=12), rnd(B, T, C, seed=13)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    def sp(t):
        return t.reshape(B, T, H, d).permute(0, 2, 1, 3)
    s = sp(qr) @ sp(kr).transpose(-1, -2) * d ** -0.5
    oref = (torch.softmax(s, -1) @ sp(vr)).permute(0, 2, 1, 3).reshape(B, T, C)
    o, lse = ops.attn_fwd(q.reshape(B * T, C), k.reshape(B * T, C), v.reshape(B * T, C), B, H, T, T, d, 1, need_lse=True)
    check("attention fwd d=40", o.reshape(B, T, C), oref, 2e-2)
    do = rnd(B, T, C, seed=14)
    oref.backward(do.float())
    dq, dk, dv = ops.attn_bwd(q.reshape(B * T, C), k.reshape(B * T, C), v.reshape(B * T, C), o, do.reshape(B * T, C), lse, B, H, T, T, d, 1)
    check("attention dq", dq.reshape(B, T, C), qr.grad, 3e-2)
    check("attention dk", dk.reshape(B, T, C), kr.grad, 3e-2)
    check("attention dv", dv.reshape(B, T, C), vr.grad, 3e-2)


def tiny_unet_and_step():
    from oracle import fair_step as fs
    from finetune_fair_diffusion_amd.step import FairnessTrainer
    om = U.oracle_models(train_unet=True, train_te=True, lora_up_std=0.05)
    # the oracle sees bf16-representable frozen weights (the reference casts the frozen models to the working dtype, :761-763)
    for name in ("unet", "vae", "text_encoder", "classifier", "eval_unet", "eval_text_encoder"):
        for p in om[name].parameters():
            if not p.requires_grad:
                p.data = p.data.to(BF).float()
    sds = {k: ({n: (t.to(BF).float() if t.is_floating_point() and k in ("unet", "vae", "clip", "clf") else t) for n, t in sd.items()})
           for k, sd in om["sds"].items()}
    pm = U.product_models(sds, dev, train_unet=True, train_te=True)
    tokens = U.tiny_tokens()
    N = 2
    with torch.no_grad():
        enc_o = fs.encode_prompts(om["text_encoder"], *tokens, N)
    x = torch.randn(2 * N, 4, 32, 32, generator=torch.Generator().manual_seed(1))
    eps_o = om["unet"](x.to(BF).float(), torch.tensor(601), encoder_hidden_states=enc_o.to(BF).float()).sample
    up = pm["unet"]
    up.prepare_timesteps([601])
    up.prepare_prompt(torch.stack([enc_o[0], enc_o[N]]).to(dev).to(BF), record=True)
    eps_p = up.forward_step(x.to(dev), 0, record=True).view(2 * N, 4, 32, 32)
    check("tiny unet eps", eps_p, eps_o, 8e-2)
    g = torch.randn(eps_o.shape, generator=torch.Generator().manual_seed(2))
    for p in om["lora_params"]:
        p.grad = None
    (eps_o * g).sum().backward()
    up.lora_bank.grad.zero_()
    up.backward_step((g * 64.0).to(dev), 64.0)
    up.finish_prompt_backward(64.0, need_denc=False)
    names = list(om["unet_lora_layers"].state_dict().keys())
    refg = torch.cat([p.grad.flatten() for p in om["unet_lora_layers"].parameters()])
    got = torch.cat([up.lora_bank.grad_view(n).flatten() for n in names])
    cos = float(F.cosine_similarity(got.cpu().double(), refg.double(), dim=0))
    print("cosine(tiny unet LoRA grads, bf16) =", cos)
    check("tiny unet LoRA grads (all)", got, refg, 2e-1)
    assert cos > 0.995
    # a complete training step (U-Net + text-encoder LoRA)
    args = U.make_args(train_unet=True, train_text_encoder=True, uncertainty_threshold=0.7)
    B, S = 4, 4
    noises = torch.randn(B, 4, 32, 32, generator=torch.Generator().manual_seed(5991))
    models_o = dict(text_encoder=om["text_encoder"], unet=om["unet"], vae=om["vae"], classifier=om["classifier"], scheduler=om["scheduler"],
                    eval_text_encoder=om["eval_text_encoder"], eval_unet=om["eval_unet"])
    for p in om["lora_params"]:
        p.grad = None
    ref = fs.fairness_step(models_o, tokens, noises, S, dict(train_GPU_batch_size=3, val_GPU_batch_size=8, uncertainty_threshold=0.7, factor2=0.2, size_face=64))
    tr = FairnessTrainer(args, pm["text_encoder"], pm["unet"], pm["vae"], pm["classifier"], pm["scheduler"], eval_text_encoder=pm["eval_text_encoder"],
                         eval_unet=pm["eval_unet"], device=dev)
    grads = {}
    tr.sync_and_update = lambda nb, apply=True: (grads.update({i: b.grad.clone() for i, b in enumerate(tr.banks)}), True)[1]
    out = tr.train_step(tokens, noises, S)
    # max over 4 x 3 x 256 x 256 pixels of a 4-step bf16 rollout against the fp32 oracle: 9.5e-2 .. 1.02e-1 of the [-1, 1] range depending on the
    # reduction order of the norms (measured before / after the single-launch GroupNorm); fp16 is at 1.3e-2.  The RMS carries the band.
    check("step: R1 images", out["images"], ref["images"], 1.5e-1)
    rms = float(((out["images"].float().cpu() - ref["images"]) ** 2).mean().sqrt())
    print(f"[bf16: step: R1 images] RMS err {rms:.3e} (band 3e-2)")
    assert rms < 3e-2
    check("step: probs", out["probs"], ref["probs"], 6e-2)
    print("targets", out["targets"].tolist(), ref["targets"].tolist(), " loss_fair", out["loss_fair"].tolist(), ref["loss_fair"].tolist())
    assert out["targets"].tolist() == ref["targets"].tolist() and out["grad_is_finite"]
    check("step: loss_fair", out["loss_fair"], ref["loss_fair"], 5e-2)
    got = torch.cat([tr.banks[0].view(n, grads[0]).flatten() for n in names])
    refg = torch.cat([p.grad.flatten() for p in om["unet_lora_layers"].parameters()])
    cos = float(F.cosine_similarity(got.cpu().double(), refg.double(), dim=0))
    print("cosine(step unet grads, bf16) =", cos)
    # measured 0.87: the bf16 images differ from the fp32 oracle's by 9.5e-2 of their range (fp16: 1.3e-2), which flips several times more
    # ReLU masks of the random-weight classifier and clamp(-1,1) masks than fp16 does (fp16: cosine 0.98); the U-Net chain itself -- no
    # discontinuity -- is pinned above at cosine 0.9998
    assert cos > 0.8


def exp3_step_d40():
    """BASELINE configs[2] logic at configs[4] precision: one complete exp-3 step (exp-3-debias-gender-race/1-main-debias.py:2016-2146: 6-logit
    head, loss = CE_gender + CE_race) in bf16 on a two-level U-Net with
    SD-v1.5's head dims (40 / 80).  Both sides train on the ORACLE's OT targets, so the band is on arithmetic, not on target flips."""
    from oracle import fair_step as fs
    from finetune_fair_diffusion_amd.fairness import EXPERIMENT_ATTRS
    from finetune_fair_diffusion_amd.step import FairnessTrainer
    ncls, attrs, cdfs, asym = EXPERIMENT_ATTRS["exp-3"]
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    om = U.oracle_models(train_unet=True, train_te=False, lora_up_std=0.05, num_classes=ncls, size="d40")
    for name in ("unet", "vae", "text_encoder", "classifier", "eval_unet"):
        for p in om[name].parameters():
            if not p.requires_grad:
                p.data = p.data.to(BF).float()
    sds = {k: ({n: (t.to(BF).float() if t.is_floating_point() and k in ("unet", "vae", "clip", "clf") else t) for n, t in sd.items()})
           for k, sd in om["sds"].items()}
    pm = U.product_models(sds, dev, train_unet=True, train_te=False, num_classes=ncls, size="d40")
    thr = 0.7
    args = U.make_args(train_unet=True, train_text_encoder=False, uncertainty_threshold=thr)
    tokens = U.tiny_tokens()
    B, S = 4, 3
    noises = torch.randn(B, 4, 32, 32, generator=torch.Generator().manual_seed(17))
    t0 = time.time()
    tg_o, img_o, _ = U.oracle_multi_targets(om, tokens, noises, S, attrs, cdfs, asym, seed=4321, thr=thr)
    assert sum(int((t != -1).sum()) for t in tg_o.values()) >= 3, tg_o
    models_o = dict(text_encoder=om["text_encoder"], unet=om["unet"], vae=om["vae"], classifier=om["classifier"], scheduler=om["scheduler"])
    for p in om["lora_params"]:
        p.grad = None
    ref = fs.fairness_step_multi(models_o, tokens, noises, S, dict(train_GPU_batch_size=3, size_face=64), attrs, tg_o)
    print(f"oracle exp-3 step on the d=40 model: {time.time() - t0:.1f} s; targets", {k: v.tolist() for k, v in tg_o.items()})
    tr = FairnessTrainer(args, pm["text_encoder"], pm["unet"], pm["vae"], pm["classifier"], pm["scheduler"], eval_unet=pm["eval_unet"],
                         experiment="exp-3", device=dev)
    tr.start_dynamic_targets = lambda per, Bn: None
    tr.finish_dynamic_targets = lambda: [(tg_o[name].clone(), torch.zeros(B)) for name, _, _ in attrs]
    grads = {}
    tr.sync_and_update = lambda nb, apply=True: (grads.update({i: b.grad.clone() for i, b in enumerate(tr.banks)}), True)[1]
    out = tr.train_step(tokens, noises, S)
    check("exp-3 d40: R1 images", out["images"], img_o, 1.5e-1)
    rms = float(((out["images"].float().cpu() - img_o) ** 2).mean().sqrt())
    print(f"[exp-3 d40: R1 images] RMS err {rms:.3e} (band 4e-2)")
    assert rms < 4e-2 and out["grad_is_finite"]
    for name, _, _ in attrs:
        check(f"exp-3 d40: loss_fair_{name}", out["loss_fair_by_attr"][name], ref["losses"][name], 8e-2)
    names = list(om["unet_lora_layers"].state_dict().keys())
    refg = torch.cat([p.grad.flatten() for p in om["unet_lora_layers"].parameters()])
    got = torch.cat([tr.banks[0].view(n, grads[0]).flatten() for n in names])
    cos = float(F.cosine_similarity(got.cpu().double(), refg.double(), dim=0))
    print("cosine(exp-3 d40 unet grads) =", cos, " norm ratio =", float(got.norm().cpu() / refg.norm()))
    assert cos > 0.80 and 0.6 < float(got.norm().cpu() / refg.norm()) < 1.6


def sd15_unet():
    from oracle import fair_step as fs
    from finetune_fair_diffusion_amd import factory
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    om = U.oracle_models(train_unet=True, train_te=False, lora_up_std=0.02, size="sd15", eval_copies=False)
    for name in ("unet", "text_encoder"):
        for p in om[name].parameters():
            if not p.requires_grad:
                p.data = p.data.to(BF).float()
    sds = {k: ({n: (t.to(BF).float() if t.is_floating_point() and k in ("unet", "vae", "clip", "clf") else t) for n, t in sd.items()})
           for k, sd in om["sds"].items()}
    from finetune_fair_diffusion_amd import weights as W
    from finetune_fair_diffusion_amd.unet import UNet2DConditionModel
    unet_p = UNet2DConditionModel(W.UNetConfig(), sds["unet"], dev)
    unet_p.add_lora(4, sds["unet_lora"])
    tokens = factory.synthetic_tokens(13, 49408)
    with torch.no_grad():
        enc = fs.encode_prompts(om["text_encoder"], *tokens, 1)
        x1 = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(1))
        t0 = time.time()
        eps_o = om["unet"](torch.cat([x1, x1]).to(BF).float(), torch.tensor(601), encoder_hidden_states=enc.to(BF).float()).sample
        print(f"oracle SD15 U-Net forward: {time.time() - t0:.1f} s")
    unet_p.prepare_timesteps([601])
    unet_p.prepare_prompt(enc.to(dev).to(BF), record=False)
    eps_p = unet_p.forward_step(x1.to(dev), 0, record=False, pair=True).view(2, 4, 64, 64)
    check("SD15 unet eps", eps_p, eps_o, 4e-2)
    rms = float((eps_p.float().cpu() - eps_o).pow(2).mean().sqrt() / eps_o.pow(2).mean().sqrt())
    print(f"SD15 unet eps rel RMS err = {rms:.3e}")
    assert rms < 2e-2


if __name__ == "__main__":
    which = sys.argv[1:] or ["kernels", "tiny", "sd15"]
    if "kernels" in which:
        kernels()
    if "tiny" in which:
        tiny_unet_and_step()
    if "sd15" in which:
        sd15_unet()
    if "exp3" in which:
        exp3_step_d40()
    print("BF16 CHECKS PASSED")
