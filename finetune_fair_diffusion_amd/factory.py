"""Builds the model set of one training run (text encoder, U-Net (+frozen copy), VAE decoder, classifier,
scheduler, trainer) on one device -- with synthetic weights when no checkpoint directory is mounted
(the build/bench environment has no network: SD-v1.5, the CLIP vocabulary and data.zip are absent).
"""
import types

import torch

from . import weights as W
from .classifier import MobileNetV3Large
from .scheduler import DPMSolverMultistepScheduler
from .step import FairnessTrainer
from .text_encoder import CLIPTextModel
from .unet import UNet2DConditionModel
from .vae import AutoencoderKL

SD15 = dict(unet=W.UNetConfig(), vae=W.VAEConfig(), clip=W.CLIPTextConfig(), clip_vision=W.CLIP_VIT_H14, dino=W.DINOV2_VITB14)
TINY = dict(unet=W.UNetConfig(block_out_channels=(64, 128, 256, 256), attention_head_dim=4, cross_attention_dim=64, sample_size=32),
            vae=W.VAEConfig(block_out_channels=(32, 64, 64, 64)),
            clip=W.CLIPTextConfig(vocab_size=1000, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2),
            clip_vision=W.ViTConfig(kind="clip", image_size=56, hidden_size=160, num_hidden_layers=2, num_attention_heads=2, intermediate_size=320,
                                    projection_dim=48, pos_grid=4),
            dino=W.ViTConfig(kind="dino", image_size=56, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                             projection_dim=0, layer_norm_eps=1e-6, pos_grid=6))


def synthetic_tokens(L=13, vocab=49408, seed=1):
    """Token ids shaped like the reference's tokenizer calls (:1007, :1020-1026): prompt = BOS, L-2 words, EOS
    (mask all ones); uncond = BOS, EOS, then EOS padding with attention_mask = [1,1,0,...]."""
    g = torch.Generator().manual_seed(seed)
    bos, eos = vocab - 2, vocab - 1
    lo, hi = min(320, vocab // 4), min(40000, vocab - 2)
    ids = torch.cat([torch.tensor([bos]), torch.randint(lo, hi, (L - 2,), generator=g), torch.tensor([eos])])
    mask = torch.ones(L, dtype=torch.long)
    uids = torch.tensor([bos] + [eos] * (L - 1))
    umask = torch.tensor([1, 1] + [0] * (L - 2))
    return ids, mask, uids, umask


def default_args(experiment="exp-1", **kw):
    from .cli import parse_args
    a = vars(parse_args([], experiment=experiment))
    a.update(kw)
    return types.SimpleNamespace(**a)


def build_trainer(args, device, cfgs=SD15, seed=0, rank=0, world_size=1, experiment="exp-1", classifier_gain=1.4, state_dicts=None,
                  frozen_copies=True, regularisers=False, lora_up_std=0.0, face_provider=None):
    """Returns (trainer, models dict).  ``state_dicts`` may carry real weights by diffusers key
    (keys 'unet','vae','clip','clf','unet_lora','te_lora'); anything missing is synthetic.
    ``frozen_copies=False`` skips the original-model replicas of R2 (inference-only consumers such as generate.py).
    ``regularisers=True`` attaches the CLIP / DINOv2 image encoders of the image-semantics loss term (keys 'clip_vision','dino';
    ``args.img_size_small`` must equal their input size) when ``weight_loss_img`` != 0, and the SFNet-20 face-feature network plus
    its feature database (keys 'face_net','face_db') when ``weight_loss_face`` != 0.
    ``lora_up_std``: fresh LoRA ``up`` matrices are ZERO, exactly like diffusers' ``LoRALinearLayer`` / ``_modify_text_encoder``
    (:798-818, :829-883), so the finetuned model equals the frozen original at step 0 and R1 == R2.  Only synthetic bench / smoke /
    test runs pass a non-zero std (the "one warm-up optimiser step" of SURVEY 8d) so that dL/d(down) is non-zero on the first step."""
    from .fairness import EXPERIMENT_ATTRS
    num_classes = EXPERIMENT_ATTRS[experiment][0]
    train_prefix = experiment == "exp-2"          # prefix-token tuning: no LoRA anywhere, both networks frozen (exp-2 :946)
    if not hasattr(args, "train_unet"):           # exp-2's CLI has neither switch
        args.train_unet = False
    if not hasattr(args, "train_text_encoder"):
        args.train_text_encoder = False
    if train_prefix and (args.train_unet or args.train_text_encoder):
        raise ValueError("exp-2 trains the prefix embedding only")
    sds = dict(state_dicts or {})
    gen = lambda shapes, s, **k: W.synthetic_state_dict(shapes, seed=seed + s, **k)  # noqa: E731
    if "unet" not in sds:
        sds["unet"] = gen(W.unet_param_shapes(cfgs["unet"]), 1)
    if "vae" not in sds:
        sds["vae"] = gen(W.vae_param_shapes(cfgs["vae"]), 2)
    if "clip" not in sds:
        sds["clip"] = gen(W.clip_param_shapes(cfgs["clip"]), 3)
    if "clf" not in sds:
        sds["clf"] = gen(W.mobilenet_param_shapes(num_classes), 4, gain=classifier_gain)
    unet = UNet2DConditionModel(cfgs["unet"], sds["unet"], device)
    # exp-2 runs R1 (debiased prompt) and R2 (plain prompt) through the SAME frozen weights; a second U-Net object (its own prompt K/V cache
    # and recorded activations) lets the two rollouts run on two streams and R3 consume R1's forward, as in the LoRA experiments
    eval_unet = UNet2DConditionModel(cfgs["unet"], sds["unet"], device) if ((args.train_unet or train_prefix) and frozen_copies) else None
    del sds["unet"]
    vae = AutoencoderKL(cfgs["vae"], sds["vae"], device)
    te = CLIPTextModel(cfgs["clip"], sds["clip"], device)
    eval_te = CLIPTextModel(cfgs["clip"], sds["clip"], device) if (args.train_text_encoder and frozen_copies) else None
    clf = MobileNetV3Large(sds["clf"], device, num_classes)
    if args.train_unet:
        bank = unet.add_lora(args.rank, sds.get("unet_lora"), seed=seed + 5)
        if "unet_lora" not in sds and lora_up_std:   # synthetic warm-up so the up matrices are non-zero
            g = torch.Generator().manual_seed(seed + 7)
            for n in bank.names:
                if ".up." in n:
                    bank.view(n).copy_((torch.randn(bank.shape(n), generator=g) * lora_up_std).to(device))
            bank.ema.copy_(bank.flat)
            unet.refresh_lora()
    if args.train_text_encoder:
        bank = te.add_lora(args.rank, sds.get("te_lora"), seed=seed + 6)
        if "te_lora" not in sds and lora_up_std:
            g = torch.Generator().manual_seed(seed + 8)
            for n in bank.names:
                if ".up." in n:
                    bank.view(n).copy_((torch.randn(bank.shape(n), generator=g) * lora_up_std).to(device))
            bank.ema.copy_(bank.flat)
            te.refresh_lora()
    import os
    import torch.distributed as _d
    if world_size > 1 or (os.environ.get("FD_FORCE_COLLECTIVES") is not None and _d.is_available() and _d.is_initialized()):  # identical LoRA init on every rank (:820-821, :848-854): one flat broadcast per bank
        import torch.distributed as dist
        for m in (unet if args.train_unet else None, te if args.train_text_encoder else None):
            if m is not None:
                dist.broadcast(m.lora_bank.flat, src=0)
                m.lora_bank.ema.copy_(m.lora_bank.flat)
                m.refresh_lora()
    prefix = None
    if train_prefix:
        from .prefix import PrefixEmbedding
        prefix = PrefixEmbedding(te, getattr(args, "train_num_tokens", 5), device, seed=seed + 13, state_dict=sds.get("prefix_embedding"))
        if world_size > 1 or (os.environ.get("FD_FORCE_COLLECTIVES") is not None and _d.is_available() and _d.is_initialized()):
            _d.broadcast(prefix.bank.flat, src=0)      # the resized embedding table is broadcast from rank 0 (:922)
            prefix.bank.ema.copy_(prefix.bank.flat)
    clip_model = dino_model = None
    if regularisers and getattr(args, "weight_loss_img", 0) != 0:
        from .vit import VisionTransformer
        for key, s in (("clip_vision", 9), ("dino", 10)):
            if key not in sds:
                sds[key] = gen(W.vit_param_shapes(cfgs[key]), s)
            if cfgs[key].image_size != args.img_size_small:
                raise ValueError(f"{key}: encoder input {cfgs[key].image_size} != --img_size_small {args.img_size_small}")
        clip_model = VisionTransformer(cfgs["clip_vision"], sds.pop("clip_vision"), device, W.CLIP_IMAGE_MEAN, W.CLIP_IMAGE_STD)
        dino_model = VisionTransformer(cfgs["dino"], sds.pop("dino"), device, W.DINO_IMAGE_MEAN, W.DINO_IMAGE_STD)
    face_net = face_db = None
    if regularisers and getattr(args, "weight_loss_face", 0) != 0:
        from .sfnet import SFNet20
        if "face_net" not in sds:
            sds["face_net"] = gen(W.sfnet20_param_shapes(in_size=args.size_aligned_face), 11)
        if "face_db" not in sds:   # FaceFeatsModel's face_feats.pkl (:80-92): here 4096 random unit vectors
            sds["face_db"] = torch.nn.functional.normalize(torch.randn(4096, 512, generator=torch.Generator().manual_seed(seed + 12)), dim=-1)
        face_net = SFNet20(sds.pop("face_net"), device, in_size=args.size_aligned_face)
        face_db = sds.pop("face_db")
    sch = DPMSolverMultistepScheduler()
    tr = FairnessTrainer(args, te, unet, vae, clf, sch, eval_text_encoder=eval_te, eval_unet=eval_unet, experiment=experiment, rank=rank,
                         world_size=world_size, device=device, clip_model=clip_model, dino_model=dino_model, face_net=face_net, face_db=face_db,
                         prefix_embedding=prefix, face_provider=face_provider)
    return tr, dict(prefix_embedding=prefix, clip_vision=clip_model, dino=dino_model, face_net=face_net, unet=unet, eval_unet=eval_unet, vae=vae, text_encoder=te, eval_text_encoder=eval_te, classifier=clf, scheduler=sch)
