"""Builds matching (oracle, product) model pairs from the SAME synthetic state dicts (test helper)."""
import copy
import types

import torch

from finetune_fair_diffusion_amd import weights as W

TINY_UNET = dict(block_out_channels=(64, 128, 256, 256), attention_head_dim=4, cross_attention_dim=64, sample_size=32)
TINY_VAE = dict(block_out_channels=(32, 64, 64, 64))
TINY_CLIP = dict(vocab_size=1000, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2)


# two-level U-Net with SD-v1.5's channel widths and head count: head dims 40 / 80 at 1024 / 256 tokens -- the smallest model whose
# self-attention is taken by the e4m3 attention kernels (T % 64 == 0, d in {40, 80, 160}; BASELINE configs[4])
D40_UNET = dict(block_out_channels=(320, 640), attention_head_dim=8, cross_attention_dim=64, sample_size=32,
                down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"))
SIZES = {"tiny": (TINY_UNET, TINY_VAE, TINY_CLIP), "sd15": ({}, {}, {}),   # "sd15": the dataclass defaults == SD-v1.5
         "d40": (D40_UNET, TINY_VAE, TINY_CLIP)}


def tiny_tokens(L=7, vocab=1000):
    ids = torch.tensor([vocab - 1] + list(range(5, 5 + L - 2)) + [vocab - 2])
    mask = torch.ones(L, dtype=torch.long)
    uids = torch.tensor([vocab - 1] + [vocab - 2] * (L - 1))
    umask = torch.tensor([1, 1] + [0] * (L - 2))
    return ids, mask, uids, umask


def make_args(**kw):
    d = dict(train_unet=True, train_text_encoder=False, rank=4, guidance_scale=7.5, train_GPU_batch_size=3, val_GPU_batch_size=8,
             uncertainty_threshold=0.2, factor1=0.2, factor2=0.2, size_face=64, learning_rate=5e-5, adam_beta1=0.9, adam_beta2=0.999,
             adam_weight_decay=1e-2, adam_epsilon=1e-8, EMA_decay=0.996, weight_loss_img=0.0, weight_loss_face=0.0, img_size_small=224,
             size_aligned_face=112, face_gender_confidence_level=0.9)
    d.update(kw)
    return types.SimpleNamespace(**d)


def synthetic_sds(rank=4, train_unet=True, train_te=False, num_classes=80, lora_up_std=0.02, seed=0, size="tiny", clf_gain=1.4):
    """The seeded synthetic state dicts both sides are built from (frozen weights fp16-representable, LoRA ``up`` ~ N(0, lora_up_std)).
    Deterministic in its arguments: the golden fixtures of tests/golden/make_oracle_step_golden.py and the GPU tests that consume
    them rebuild identical weights from identical seeds."""
    UK, VK, CK = SIZES[size]
    sds = dict(
        unet=W.synthetic_state_dict(W.unet_param_shapes(W.UNetConfig(**UK)), seed=seed + 1),
        vae=W.synthetic_state_dict(W.vae_param_shapes(W.VAEConfig(**VK)), seed=seed + 2),
        clip=W.synthetic_state_dict(W.clip_param_shapes(W.CLIPTextConfig(**CK)), seed=seed + 3),
        clf=W.synthetic_state_dict(W.mobilenet_param_shapes(num_classes), seed=seed + 4, gain=clf_gain),
    )
    # the frozen models are cast to fp16 in the reference (:761-763): make the oracle see fp16-representable weights
    for k in ("unet", "vae", "clip", "clf"):
        for n, t in sds[k].items():
            if t.is_floating_point():
                sds[k][n] = t.half().float()
    g = torch.Generator().manual_seed(seed + 10)
    if train_unet:
        sd = W.synthetic_state_dict(W.unet_lora_param_shapes(W.UNetConfig(**UK), rank), seed=seed + 5)
        for n in sd:
            if ".up." in n:
                sd[n] = torch.randn(sd[n].shape, generator=g) * lora_up_std
        sds["unet_lora"] = sd
    if train_te:
        sd = W.synthetic_state_dict(W.clip_lora_param_shapes(W.CLIPTextConfig(**CK), rank), seed=seed + 6)
        for n in sd:
            if ".up." in n:
                sd[n] = torch.randn(sd[n].shape, generator=g) * lora_up_std
        sds["te_lora"] = sd
    return sds


def oracle_models(rank=4, train_unet=True, train_te=False, num_classes=80, lora_up_std=0.02, seed=0, size="tiny", eval_copies=True,
                  clf_gain=1.4):
    """Returns dict of oracle modules + the state dicts they were loaded from.  ``size``: "tiny" or "sd15" (SD-v1.5 shapes)."""
    from oracle import nn_clip, nn_mobilenet, nn_unet, nn_vae
    from oracle.dpm_solver import DPMSolverMultistepScheduler
    UK, VK, CK = SIZES[size]
    ucfg, vcfg, ccfg = nn_unet.UNetConfig(**UK), nn_vae.VAEConfig(**VK), nn_clip.CLIPTextConfig(**CK)
    sds = synthetic_sds(rank, train_unet, train_te, num_classes, lora_up_std, seed, size, clf_gain)
    unet = nn_unet.UNet2DConditionModel(ucfg)
    unet.load_state_dict(sds["unet"], strict=True)
    eval_unet = copy.deepcopy(unet) if (eval_copies and train_unet) or size == "tiny" else unet
    vae = nn_vae.AutoencoderKLDecoder(vcfg)
    vae.load_state_dict(sds["vae"], strict=True)
    te = nn_clip.CLIPTextModel(ccfg)
    te.load_state_dict(sds["clip"], strict=True)
    eval_te = copy.deepcopy(te) if (eval_copies and train_te) or size == "tiny" else te
    clf = nn_mobilenet.MobileNetV3Large(num_classes).eval()
    clf.load_state_dict(sds["clf"], strict=True)
    for m in {id(m): m for m in (unet, eval_unet, vae, te, eval_te, clf)}.values():
        m.requires_grad_(False)
    out = dict(unet=unet, eval_unet=eval_unet, vae=vae, text_encoder=te, eval_text_encoder=eval_te, classifier=clf,
               scheduler=DPMSolverMultistepScheduler(), sds=sds, lora_params=[])
    if train_unet:
        layers = nn_unet.make_unet_lora(unet, rank)
        layers.load_named(sds["unet_lora"])
        out["unet_lora_layers"] = layers
        for p in layers.parameters():
            p.requires_grad_(True)
        out["lora_params"] += list(layers.parameters())
    if train_te:
        params = nn_clip.modify_text_encoder(te, rank)
        missing, unexpected = te.load_state_dict(sds["te_lora"], strict=False)
        assert not unexpected, unexpected
        for p in params:
            p.requires_grad_(True)
        out["lora_params"] += params
        out["te_lora_named"] = {n: p for n, p in te.named_parameters() if "lora_linear_layer" in n}
    return out


def product_models(sds, dev, rank=4, train_unet=True, train_te=False, num_classes=80, size="tiny", eval_copies=True):
    from finetune_fair_diffusion_amd.classifier import MobileNetV3Large
    from finetune_fair_diffusion_amd.scheduler import DPMSolverMultistepScheduler
    from finetune_fair_diffusion_amd.text_encoder import CLIPTextModel
    from finetune_fair_diffusion_amd.unet import UNet2DConditionModel
    from finetune_fair_diffusion_amd.vae import AutoencoderKL
    UK, VK, CK = SIZES[size]
    ucfg, vcfg, ccfg = W.UNetConfig(**UK), W.VAEConfig(**VK), W.CLIPTextConfig(**CK)
    unet = UNet2DConditionModel(ucfg, sds["unet"], dev)
    eval_unet = UNet2DConditionModel(ucfg, sds["unet"], dev) if (train_unet and eval_copies) else None
    vae = AutoencoderKL(vcfg, sds["vae"], dev)
    te = CLIPTextModel(ccfg, sds["clip"], dev)
    eval_te = CLIPTextModel(ccfg, sds["clip"], dev) if (train_te and eval_copies) else None
    clf = MobileNetV3Large(sds["clf"], dev, num_classes)
    if train_unet:
        unet.add_lora(rank, sds["unet_lora"])
    if train_te:
        # oracle keys carry the PatchedLoraProjection naming of the export format
        te.add_lora(rank, sds["te_lora"])
    return dict(unet=unet, eval_unet=eval_unet, vae=vae, text_encoder=te, eval_text_encoder=eval_te, classifier=clf,
                scheduler=DPMSolverMultistepScheduler())


def oracle_multi_targets(om, tokens, noises, S, attrs, cdfs, asym, seed, thr, size_face=64):
    """The oracle's own R1 -> classifier -> Monte-Carlo OT (LP solver) targets (exp-3 :2016-2025, exp-4 :2157-2170)."""
    from oracle import fair_step as fs
    with torch.no_grad():
        img = fs.generate_image_no_gradient(tokens, noises, S, om["text_encoder"], om["unet"], om["vae"], om["scheduler"])
        ind, _, chips = fs.SyntheticFaceProvider(size_face)(img)
        lo = om["classifier"](chips[ind])
    probs = []
    for _, c0, k in attrs:
        p = torch.ones(noises.shape[0], k) * (-1)
        p[ind] = torch.softmax(lo[:, c0:c0 + k], dim=-1)
        probs.append(p)
    res, tp = fs.generate_dynamic_targets_multi(probs, cdfs, 100, torch.Generator().manual_seed(seed), asym)
    out = {}
    for (name, _, _), (t, u) in zip(attrs, res):
        t = t.clone()
        t[u > thr] = -1
        out[name] = t
    return out, img, probs


class SmoothHeadProduct:
    """Test double with the classifier's contract (num_classes, forward(chips, record), backward(d_logits, gscale), _ctx) but NO
    discontinuity: logits = W2 hardswish(W1 vec(chips) + b1) + b2, on the product's own kernels through the C-ABI (MFMA GEMMs,
    fd_act_fwd/bwd).  With it the only non-smooth op left between the LoRA weights and the loss is images.clamp(-1, 1)."""

    def __init__(self, w1, b1, w2, b2, dev):
        self.w1, self.w2 = w1.to(dev).half().contiguous(), w2.to(dev).half().contiguous()
        self.w1T, self.w2T = self.w1.t().contiguous(), self.w2.t().contiguous()
        self.b1, self.b2 = b1.to(dev).float().contiguous(), b2.to(dev).float().contiguous()
        self.num_classes, self._ctx = w2.shape[0], None

    def forward(self, chips, record=False):
        from finetune_fair_diffusion_amd import ops
        n = chips.shape[0]
        x = torch.zeros(((n + 7) // 8 * 8, self.w1.shape[1]), dtype=torch.float16, device=chips.device)
        x[:n] = chips.reshape(n, -1)
        z1 = ops.gemm(x, self.w1, bias=self.b1)
        h = ops.act_fwd(z1, "hardswish")
        logits = ops.gemm(h, self.w2, bias=self.b2, out_dtype=torch.float32)
        if record:
            self._ctx = dict(z1=z1, n=n, shape=chips.shape)
        return logits[:n]

    def backward(self, d_logits, gscale):
        from finetune_fair_diffusion_amd import ops
        c = self._ctx
        d = torch.zeros((c["z1"].shape[0], self.num_classes), dtype=torch.float32, device=d_logits.device)
        d[:c["n"]] = d_logits
        dh = ops.gemm(ops.to_f16(d.contiguous(), gscale), self.w2T)
        dz = ops.act_bwd(c["z1"], dh, "hardswish")
        dx = ops.gemm(dz, self.w1T, out_dtype=torch.float32, alpha=1.0 / gscale)
        self._ctx = None
        return dx[:c["n"]].reshape(c["shape"]).contiguous()
