"""Golden vectors of the CPU oracle's SD-v1.5-SIZE runs ("latents after each scheduler step, loss_fair, LoRA grads of
3 named tensors, dynamic targets / uncertainties ... committed as small fixtures and are what the MI355X run is checked against").

CPU only.  ``latents`` (no-grad rollout, ~9 GB of host RAM, 2.5 min on 8 cores) runs in the build container; ``cfg0`` and ``smooth`` hold
the oracle's autograd graph of the U-Net at SD-v1.5 size (more than the build container's 62 GB) and were generated on the host CPU of
a GPU box with this same script (scratch/r03_passes.sh a copies the files back):

    python tests/golden/make_oracle_step_golden.py [cfg0] [latents] [smooth] [smooth_te] [cfg0_b8] [loss_seeds] [loss_seeds_more] [loss_seeds_fp16]

(round 5: ``smooth_te``, ``cfg0_b8`` and ``loss_seeds`` were generated the same way, scratch/r05_passes.sh d.)

The inputs are fully synthetic and seeded (tests/util_models.py: weights.synthetic_state_dict seeds, factory.synthetic_tokens, CPU-drawn
noise from torch.manual_seed(5991)), so the GPU box rebuilds the SAME product models from the same seeds and compares against the vectors
stored here without re-running the oracle.  Test infrastructure only (like everything under oracle/): nothing in the product reads it.

  oracle_sd15_cfg0_b2_s4_te_lora.npz   BASELINE configs[0]: exp-1, batch 2, 4 denoising steps, LoRA r=4 on the text encoder only --
                                       the COMPLETE step (exp-1-debias-gender/1-main-debias.py:1746-2029): R1 latents after every
                                       scheduler step (:1131), probabilities, dynamic targets + uncertainties (:1403-1447), loss_fair
                                       (:1912-1916), three named text-encoder LoRA gradients, a seeded sample of the flat gradient
  oracle_sd15_r1_latents_b1_s20.npz    BASELINE configs[1] rollout length: LoRA r=4 on the U-Net, batch 1, 20 steps -- the no-grad CFG
                                       rollout (:1038-1056): latents after each of the 20 scheduler steps
  oracle_sd15_smooth_head_b2_s2.npz    the complete step with U-Net LoRA and a classifier double WITHOUT discontinuities (two linear
                                       layers + hardswish), batch 2, 2 steps: loss, probabilities, targets, three named U-Net LoRA
                                       gradients, a seeded 65536-entry sample of the flat LoRA gradient and its norm (pins the backward
                                       chain end to end at full size)
  oracle_sd15_smooth_head_te_lora_b2_s4.npz   round 5: BASELINE configs[0]'s trainable set -- LoRA r=4 on the TEXT ENCODER only -- with the
                                       smooth classifier double, batch 2, 4 steps: d prompt_embeds accumulated over the 4 x 16 cross-attention K / V
                                       projections of the CFG pair + the CLIP text backward, which no tight gate pinned before
  oracle_sd15_cfg0_b8_s4_te_lora.npz   round 5 (item 2b): the cfg0 step with EIGHT images and the real ReLU / hard-swish classifier (three micro-batches of
                                       3, 3, 2 as in the reference): more terms average the mask flips that make the two-image gradient cosine chaotic
  oracle_sd15_loss_seeds_b2_s2.npz     round 5 (item 2c): loss_fair / probabilities / targets of the forward half of the step (no gradient) for eight
                                       noise seeds, U-Net LoRA r=4, batch 2, 2 steps: mean |error| and BIAS of the product's loss over seeds
                                       (round 6, ``loss_seeds_more``: 32 further seeds APPENDED -- 64 seeds = 128 loss terms, the first 32 rows untouched)
  oracle_sd15_loss_seeds_fp16_b2_s2.npz   round 6: the same forward for the first eight seeds with the oracle's arithmetic ROUNDED TO FP16
                                       the way the reference's fp16 models round (weights cast once, :761-763; every module's output rounded to fp16; latents and
                                       prompt embeddings in fp16; fp32 accumulation inside an op, as the fp16 GPU kernels of torch do) -- the denominator of
                                       the product-vs-fp32 loss error: how far an fp16 run of the REFERENCE's own arithmetic sits from the fp32 oracle
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import util_models as U  # noqa: E402
from finetune_fair_diffusion_amd import factory  # noqa: E402  (synthetic_tokens only: host-side helper, no HIP)
from oracle import fair_step as fs  # noqa: E402

L = 13
NOISE_SEED = 5991
UNET_NAMED = ("down_blocks.0.attentions.0.transformer_blocks.0.attn1.processor.to_q_lora.up.weight",
              "mid_block.attentions.0.transformer_blocks.0.attn2.processor.to_k_lora.down.weight",
              "up_blocks.3.attentions.2.transformer_blocks.0.attn2.processor.to_out_lora.up.weight")
SAMPLE_SEED, SAMPLE_N = 20260, 65536


def smooth_head_weights(size_face=224, hidden=256, classes=80, seed=99):
    """The classifier double of tests/test_engine_gpu.py::_SmoothHeadProduct at the SD-v1.5 face-chip size."""
    g = torch.Generator().manual_seed(seed)
    K = 3 * size_face * size_face
    w1 = (torch.randn(hidden, K, generator=g) * (2.0 / K ** 0.5)).half().float()
    b1 = torch.randn(hidden, generator=g) * 0.1
    w2 = (torch.randn(classes, hidden, generator=g) * (2.0 / hidden ** 0.5)).half().float()
    b2 = torch.randn(classes, generator=g) * 0.1
    return w1, b1, w2, b2


def smooth_head_module(w1, b1, w2, b2):
    K, Hd, C = w1.shape[1], w1.shape[0], w2.shape[0]
    m = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(K, Hd), torch.nn.Hardswish(), torch.nn.Linear(Hd, C)).requires_grad_(False)
    m[1].weight.copy_(w1); m[1].bias.copy_(b1); m[3].weight.copy_(w2); m[3].bias.copy_(b2)
    return m


def grad_sample_index(n):
    return torch.randperm(n, generator=torch.Generator().manual_seed(SAMPLE_SEED))[:min(SAMPLE_N, n)]


def unet_models():
    return U.oracle_models(train_unet=True, train_te=False, lora_up_std=0.02, size="sd15", eval_copies=False)   # == tests/test_fullsize_gpu.py::full


class Frozen:
    """R2's frozen original U-Net (:1844-1858): the same module with its LoRA processors detached for the duration of a call."""

    def __init__(self, unet):
        self.u = unet

    def __call__(self, *a, **k):
        procs = dict(self.u.attn_processors)
        self.u.set_attn_processor({n: None for n in procs})
        try:
            return self.u(*a, **k)
        finally:
            self.u.set_attn_processor(procs)


def make_latents(om):
    tokens = factory.synthetic_tokens(L, 49408)
    noises = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(NOISE_SEED))
    trace = []
    t0 = time.time()
    with torch.no_grad():
        img = fs.generate_image_no_gradient(tokens, noises, 20, om["text_encoder"], om["unet"], om["vae"], om["scheduler"], 7.5, trace=trace)
    lat = torch.stack(trace)                # [20, 1, 4, 64, 64]
    print(f"latents: 20-step rollout in {time.time() - t0:.0f} s; |x_final|max = {float(lat[-1].abs().max()):.3f}")
    np.savez_compressed(os.path.join(HERE, "oracle_sd15_r1_latents_b1_s20.npz"), latents=lat.numpy().astype(np.float16),
                        image_mean=np.float32(img.mean()), image_abs_mean=np.float32(img.abs().mean()),
                        image_8x8=torch.nn.functional.avg_pool2d(img, 64).numpy().astype(np.float32))


def make_smooth(om):
    tokens = factory.synthetic_tokens(L, 49408)
    B, S = 2, 2
    noises = torch.randn(B, 4, 64, 64, generator=torch.Generator().manual_seed(NOISE_SEED))
    head = smooth_head_module(*smooth_head_weights())
    models = dict(text_encoder=om["text_encoder"], unet=om["unet"], vae=om["vae"], classifier=head, scheduler=om["scheduler"],
                  eval_text_encoder=om["text_encoder"], eval_unet=Frozen(om["unet"]))
    for p in om["lora_params"]:
        p.grad = None
    t0 = time.time()
    ref = fs.fairness_step(models, tokens, noises, S, dict(train_GPU_batch_size=3, val_GPU_batch_size=8, uncertainty_threshold=0.7, factor2=0.2,
                                                           size_face=224))
    print(f"smooth head: full step B={B} S={S} in {time.time() - t0:.0f} s; targets {ref['targets'].tolist()} loss {ref['loss_fair'].tolist()}")
    names = list(om["unet_lora_layers"].state_dict().keys())
    flat = torch.cat([p.grad.flatten() for p in om["unet_lora_layers"].parameters()])
    named = dict(zip(names, om["unet_lora_layers"].parameters()))
    idx = grad_sample_index(flat.numel())
    np.savez_compressed(os.path.join(HERE, "oracle_sd15_smooth_head_b2_s2.npz"),
                        probs=ref["probs"].numpy(), targets=ref["targets"].numpy(), uncertainty=ref["uncertainty"].numpy(),
                        loss_fair=ref["loss_fair"].numpy(), grad_norm=np.float64(flat.double().norm()), grad_sample=flat[idx].numpy(),
                        grad_absmax=np.float32(flat.abs().max()),
                        latents=torch.stack(ref["latents_trace"]).numpy().astype(np.float16),
                        **{"grad::" + n: named[n].grad.numpy() for n in UNET_NAMED})


def make_cfg0():
    om = U.oracle_models(rank=4, train_unet=False, train_te=True, lora_up_std=0.01, size="sd15", eval_copies=True)
    tokens = factory.synthetic_tokens(L, 49408)
    B, S = 2, 4
    noises = torch.randn(B, 4, 64, 64, generator=torch.Generator().manual_seed(NOISE_SEED))
    models = dict(text_encoder=om["text_encoder"], unet=om["unet"], vae=om["vae"], classifier=om["classifier"], scheduler=om["scheduler"],
                  eval_text_encoder=om["eval_text_encoder"], eval_unet=om["unet"])
    for p in om["lora_params"]:
        p.grad = None
    t0 = time.time()
    ref = fs.fairness_step(models, tokens, noises, S, dict(train_GPU_batch_size=3, val_GPU_batch_size=8, uncertainty_threshold=0.6, factor2=0.2,
                                                           size_face=224))
    print(f"cfg0: full step B={B} S={S} (TE LoRA) in {time.time() - t0:.0f} s; targets {ref['targets'].tolist()} loss {ref['loss_fair'].tolist()}")
    names = list(om["te_lora_named"].keys())
    flat = torch.cat([om["te_lora_named"][n].grad.flatten() for n in names])
    pick = [names[0], names[len(names) // 2], names[-1]]
    idx = grad_sample_index(flat.numel())
    np.savez_compressed(os.path.join(HERE, "oracle_sd15_cfg0_b2_s4_te_lora.npz"),
                        latents=torch.stack(ref["latents_trace"]).numpy().astype(np.float16),
                        probs=ref["probs"].numpy(), probs_ori=ref["probs_ori"].numpy(), targets=ref["targets"].numpy(),
                        uncertainty=ref["uncertainty"].numpy(), loss_fair=ref["loss_fair"].numpy(),
                        grad_norm=np.float64(flat.double().norm()), grad_sample=flat[idx].numpy(), grad_absmax=np.float32(flat.abs().max()),
                        named=np.array(pick), **{"grad::" + n: om["te_lora_named"][n].grad.numpy() for n in pick})


def _te_models():
    return U.oracle_models(rank=4, train_unet=False, train_te=True, lora_up_std=0.01, size="sd15", eval_copies=True)


def _save_te_step(fname, om, ref, extra=None):
    names = list(om["te_lora_named"].keys())
    flat = torch.cat([om["te_lora_named"][n].grad.flatten() for n in names])
    pick = [names[0], names[len(names) // 2], names[-1]]
    idx = grad_sample_index(flat.numel())
    np.savez_compressed(os.path.join(HERE, fname),
                        latents=torch.stack(ref["latents_trace"]).numpy().astype(np.float16),
                        probs=ref["probs"].numpy(), probs_ori=ref["probs_ori"].numpy(), targets=ref["targets"].numpy(),
                        uncertainty=ref["uncertainty"].numpy(), loss_fair=ref["loss_fair"].numpy(),
                        grad_norm=np.float64(flat.double().norm()), grad_sample=flat[idx].numpy(), grad_absmax=np.float32(flat.abs().max()),
                        named=np.array(pick), **{"grad::" + n: om["te_lora_named"][n].grad.numpy() for n in pick}, **(extra or {}))


def make_smooth_te():
    """Text-encoder LoRA with the smooth classifier double (B = 2, S = 4): the tight pin of the d prompt_embeds path."""
    om = _te_models()
    tokens = factory.synthetic_tokens(L, 49408)
    B, S = 2, 4
    noises = torch.randn(B, 4, 64, 64, generator=torch.Generator().manual_seed(NOISE_SEED))
    head = smooth_head_module(*smooth_head_weights())
    models = dict(text_encoder=om["text_encoder"], unet=om["unet"], vae=om["vae"], classifier=head, scheduler=om["scheduler"],
                  eval_text_encoder=om["eval_text_encoder"], eval_unet=om["unet"])
    for p in om["lora_params"]:
        p.grad = None
    t0 = time.time()
    ref = fs.fairness_step(models, tokens, noises, S, dict(train_GPU_batch_size=3, val_GPU_batch_size=8, uncertainty_threshold=0.7, factor2=0.2, size_face=224))
    print(f"smooth head, TE LoRA: full step B={B} S={S} in {time.time() - t0:.0f} s; targets {ref['targets'].tolist()} loss {ref['loss_fair'].tolist()}")
    _save_te_step("oracle_sd15_smooth_head_te_lora_b2_s4.npz", om, ref)


CFG0_B8_SEED = 7331


def make_cfg0_b8():
    """cfg0 (TE LoRA, real classifier) with eight images in the reference's micro-batches of three."""
    om = _te_models()
    tokens = factory.synthetic_tokens(L, 49408)
    B, S = 8, 4
    noises = torch.randn(B, 4, 64, 64, generator=torch.Generator().manual_seed(CFG0_B8_SEED))
    models = dict(text_encoder=om["text_encoder"], unet=om["unet"], vae=om["vae"], classifier=om["classifier"], scheduler=om["scheduler"],
                  eval_text_encoder=om["eval_text_encoder"], eval_unet=om["unet"])
    for p in om["lora_params"]:
        p.grad = None
    t0 = time.time()
    ref = fs.fairness_step(models, tokens, noises, S, dict(train_GPU_batch_size=3, val_GPU_batch_size=8, uncertainty_threshold=0.6, factor2=0.2, size_face=224))
    print(f"cfg0 B=8: full step S={S} (TE LoRA) in {time.time() - t0:.0f} s; targets {ref['targets'].tolist()} loss {ref['loss_fair'].tolist()}")
    _save_te_step("oracle_sd15_cfg0_b8_s4_te_lora.npz", om, ref, dict(n_backward=np.int32(ref["N_backward"])))


LOSS_SEEDS = (101, 202, 303, 404, 505, 606, 707, 808) + tuple(909 + 101 * i for i in range(24))      # 32 seeds = 64 loss terms (round 5: eight seeds could not
# tell a 1.2-sigma draw of the mean from bias)


def make_loss_seeds(om):
    """Forward half of the step (no gradient) for 32 noise seeds: the loss the product's fp16 forward must reproduce without bias."""
    tokens = factory.synthetic_tokens(L, 49408)
    B, S = 2, 2
    models = dict(text_encoder=om["text_encoder"], unet=om["unet"], vae=om["vae"], classifier=om["classifier"], scheduler=om["scheduler"],
                  eval_text_encoder=om["text_encoder"], eval_unet=Frozen(om["unet"]))
    rows = []
    t0 = time.time()
    te, unet, vae, clf, sch = (models[k] for k in ("text_encoder", "unet", "vae", "classifier", "scheduler"))
    faces = fs.SyntheticFaceProvider(224)
    for seed in LOSS_SEEDS:
        noises = torch.randn(B, 4, 64, 64, generator=torch.Generator().manual_seed(seed))
        # the forward half of fairness_step (:1746-1916) without the autograd graph: R1 images -> face chips -> probabilities -> dynamic targets ->
        # cross-entropy of the faces that have a target (the with-gradient rollout evaluates the same function on the same inputs)
        with torch.no_grad():
            images = fs.generate_image_no_gradient(tokens, noises, S, te, unet, vae, sch, 7.5)
            ind, boxes, chips = faces(images)
            preds, probs, logits = fs.get_face_gender(clf, chips, selector=ind)
            targets, unc = fs.generate_dynamic_targets(probs, w_uncertainty=True)
            targets[unc > 0.7] = -1
            lf = torch.ones(B) * (-1)
            w = ((ind == True) * (targets != -1)).nonzero().view([-1])  # noqa: E712
            lf[w] = torch.nn.functional.cross_entropy(logits[w], targets[w], reduction="none")
        ref = dict(loss_fair=lf, probs=probs, targets=targets, uncertainty=unc)
        rows.append(ref)
        print(f"loss seeds: seed {seed}: targets {ref['targets'].tolist()} loss {ref['loss_fair'].tolist()}  ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(os.path.join(HERE, "oracle_sd15_loss_seeds_b2_s2.npz"), seeds=np.array(LOSS_SEEDS),
                        loss_fair=np.stack([r["loss_fair"].numpy() for r in rows]), probs=np.stack([r["probs"].numpy() for r in rows]),
                        targets=np.stack([r["targets"].numpy() for r in rows]), uncertainty=np.stack([r["uncertainty"].numpy() for r in rows]))


LOSS_SEEDS_MORE = tuple(4243 + 97 * i for i in range(32))      # round 6: 32 more (128 terms put a 3e-4 gate on the mean at >= 2 sigma)


def _loss_forward(models, tokens, noises, S, faces, dtype=torch.float32):
    """The forward half of fairness_step (:1746-1916) without the autograd graph -- the statement make_loss_seeds stores."""
    te, unet, vae, clf, sch = (models[k] for k in ("text_encoder", "unet", "vae", "classifier", "scheduler"))
    B = noises.shape[0]
    with torch.no_grad():
        images = fs.generate_image_no_gradient(tokens, noises, S, te, unet, vae, sch, 7.5, dtype=dtype)
        ind, boxes, chips = faces(images)
        preds, probs, logits = fs.get_face_gender(clf, chips, selector=ind)
        targets, unc = fs.generate_dynamic_targets(probs, w_uncertainty=True)
        targets[unc > 0.7] = -1
        lf = torch.ones(B) * (-1)
        w = ((ind == True) * (targets != -1)).nonzero().view([-1])  # noqa: E712
        lf[w] = torch.nn.functional.cross_entropy(logits[w], targets[w], reduction="none")
    return dict(loss_fair=lf, probs=probs, targets=targets, uncertainty=unc)


def make_loss_seeds_more(om):
    """Appends LOSS_SEEDS_MORE to oracle_sd15_loss_seeds_b2_s2.npz (rows 0..31 are copied as they are)."""
    path = os.path.join(HERE, "oracle_sd15_loss_seeds_b2_s2.npz")
    g = dict(np.load(path))
    assert g["seeds"].tolist() == list(LOSS_SEEDS), "the file already holds other seeds"
    tokens = factory.synthetic_tokens(L, 49408)
    models = dict(text_encoder=om["text_encoder"], unet=om["unet"], vae=om["vae"], classifier=om["classifier"], scheduler=om["scheduler"])
    faces = fs.SyntheticFaceProvider(224)
    rows, t0 = [], time.time()
    for seed in LOSS_SEEDS_MORE:
        rows.append(_loss_forward(models, tokens, torch.randn(2, 4, 64, 64, generator=torch.Generator().manual_seed(seed)), 2, faces))
        print(f"loss seeds (more): seed {seed}: targets {rows[-1]['targets'].tolist()} loss {rows[-1]['loss_fair'].tolist()}  ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(path, seeds=np.array(LOSS_SEEDS + LOSS_SEEDS_MORE),
                        **{k: np.concatenate([g[k], np.stack([r[k].numpy() for r in rows])]) for k in ("loss_fair", "probs", "targets", "uncertainty")})


class Fp16Rounding:
    """The oracle's modules with fp16 ROUNDING and fp32 arithmetic: parameters rounded once (the reference casts the frozen models wholesale, :761-763; the fp32
    LoRA parameters are left alone, :815), the output of every leaf module rounded to fp16 (what an fp16 kernel with fp32 accumulation returns), restored on exit."""

    def __init__(self, *modules):
        self.modules = modules

    def __enter__(self):
        self.saved, self.hooks = [], []

        def rnd(_m, _i, out):
            if isinstance(out, torch.Tensor) and out.dtype == torch.float32:
                return out.half().float()
            return out
        for mod in self.modules:
            for n, p in mod.named_parameters():
                if "lora" in n:
                    continue
                self.saved.append((p, p.data.clone()))
                p.data = p.data.half().float()
            for m in mod.modules():
                if not list(m.children()):
                    self.hooks.append(m.register_forward_hook(rnd))
        return self

    def __exit__(self, *a):
        for h in self.hooks:
            h.remove()
        for p, d in self.saved:
            p.data = d


def make_loss_seeds_fp16(om, n=8):
    tokens = factory.synthetic_tokens(L, 49408)
    models = dict(text_encoder=om["text_encoder"], unet=om["unet"], vae=om["vae"], classifier=om["classifier"], scheduler=om["scheduler"])
    faces = fs.SyntheticFaceProvider(224)
    rows, t0 = [], time.time()

    class HalfIO:       # latents / prompt embeddings enter the U-Net in fp16 (dtype=float16 in the reference's rollout) and eps leaves it in fp16
        def __init__(self, u):
            self.u = u

        def __call__(self, x, t, encoder_hidden_states=None):
            r = self.u(x.half().float(), t, encoder_hidden_states=encoder_hidden_states.half().float())
            r.sample = r.sample.half().float()
            return r
    with Fp16Rounding(om["text_encoder"], om["unet"], om["vae"], om["classifier"]):
        m16 = dict(models, unet=HalfIO(om["unet"]))
        for seed in LOSS_SEEDS[:n]:
            rows.append(_loss_forward(m16, tokens, torch.randn(2, 4, 64, 64, generator=torch.Generator().manual_seed(seed)), 2, faces))
            print(f"loss seeds (fp16 rounding): seed {seed}: targets {rows[-1]['targets'].tolist()} loss {rows[-1]['loss_fair'].tolist()}  ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(os.path.join(HERE, "oracle_sd15_loss_seeds_fp16_b2_s2.npz"), seeds=np.array(LOSS_SEEDS[:n]),
                        **{k: np.stack([r[k].numpy() for r in rows]) for k in ("loss_fair", "probs", "targets", "uncertainty")})


if __name__ == "__main__":
    what = sys.argv[1:] or ["cfg0", "latents", "smooth"]
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    if "cfg0" in what:
        make_cfg0()
    if "smooth_te" in what:
        make_smooth_te()
    if "cfg0_b8" in what:
        make_cfg0_b8()
    if {"latents", "smooth", "loss_seeds", "loss_seeds_more", "loss_seeds_fp16"} & set(what):
        om = unet_models()
        if "latents" in what:
            make_latents(om)
        if "smooth" in what:
            make_smooth(om)
        if "loss_seeds" in what:
            make_loss_seeds(om)
        if "loss_seeds_fp16" in what:
            make_loss_seeds_fp16(om)
        if "loss_seeds_more" in what:
            make_loss_seeds_more(om)
