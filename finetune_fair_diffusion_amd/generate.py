"""Inference consumer of the exported LoRA files -- the reference's ``gen-images.py`` (:112-175 ``generate_image``,
:345-450 flags, :493-612 main) on the no-grad rollout of this package (the same kernels as R1/R2 of the training step).

    python -m finetune_fair_diffusion_amd.generate --prompts_path prompts.json --save_dir out \\
        --load_unet_lora_from <ckpt>_exported/unet_lora_EMA.pth --load_text_encoder_lora_from <ckpt>_exported/text_encoder_lora_EMA.pth

Same flags and defaults, same per-image noise seeding ``torch.manual_seed(random_seed + hash(prompt) + i)`` (:537, which
in the reference, as here, depends on PYTHONHASHSEED), same skip-existing-files resume behaviour and output tree
``save_dir/prompt_{i}/img_{j}.jpg``.  ``--load_prefix_embedding_from`` (exp-2's learned prefix tokens, gen-images.py:272-343, :523-538): the
``number_prefix_tokens`` placeholder tokens are prepended to every prompt and their embeddings replaced by the loaded ones
(``prefix_tokens`` / ``generate_image(..., prefix=)``); the uncond branch follows ``StableDiffusionPipeline._encode_prompt`` (no padding mask).
"""
import argparse
import json
import math
import os

import torch

from .factory import SD15, TINY, build_trainer, default_args


def parse_args(input_args=None):
    p = argparse.ArgumentParser(description="Script to generate images with (debiased) Stable Diffusion.")
    a = p.add_argument
    a("--pretrained_model_name_or_path", type=str, default="runwayml/stable-diffusion-v1-5")
    a("--load_text_encoder_lora_from", type=str, default=None)
    a("--load_unet_lora_from", type=str, default=None)
    a("--load_prefix_embedding_from", type=str, default=None)
    a("--number_prefix_tokens", type=int, default=5)
    a("--gpu_id", type=int, default=0)
    a("--prompts_path", type=str, required=True)
    a("--num_imgs_per_prompt", type=int, default=64)
    a("--save_dir", type=str, default=None, required=True)
    a("--random_seed", type=int, default=1997)
    a("--resume_from_checkpoint", type=str, default=None)
    a("--mixed_precision", type=str, default="fp16", choices=["no", "fp16", "bf16"])
    a("--rank", type=int, default=50)
    a("--guidance_scale", type=float, default=7.5)
    a("--num_denoising_steps", type=int, default=30)
    a("--batch_size", type=int, default=10)
    a("--synthetic", action="store_true", default=False, help="(build addition) synthetic base weights / hash tokenizer")
    return p.parse_args(input_args) if input_args is not None else p.parse_args()


def to_uint8_hwc(images):
    """``transforms.ToPILImage()(img*0.5+0.5)`` (:607-608): float CHW in [0,1] -> ``mul(255).byte()`` (truncating)."""
    x = images.float() * 0.5 + 0.5
    return x.mul(255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().cpu().numpy()


def prefix_tokens(tokens, n, vocab_size):
    """The token tuple of ``"".join(prefix_tokens) + prompt`` (gen-images.py:524-526): n placeholder ids (``expand_tokenizer`` appends them
    to the vocabulary: vocab_size .. vocab_size + n - 1) between BOS and the prompt's words; the uncond sequence is the pipeline's
    ``[""]`` padded to the new length and -- unlike ``generate_image`` -- evaluated WITHOUT a padding mask
    (``_encode_prompt``: ``attention_mask=None`` because SD-v1.5's text encoder config has no ``use_attention_mask``)."""
    pid, pm, uid, um = tokens
    ids = torch.cat([pid[:1], torch.arange(vocab_size, vocab_size + n, dtype=pid.dtype), pid[1:]])
    L = ids.shape[0]
    uids = torch.cat([uid[:1], uid[1:2].expand(L - 1)])
    return ids, torch.ones(L, dtype=pm.dtype), uids, torch.ones(L, dtype=um.dtype)


def load_prefix_embedding(path, n):
    """``FairEmbeddings`` state dict (gen-images.py:537-538, ``strict=False``): rows 1..n of ``token_embedding.weight`` are the prefix vectors."""
    sd = torch.load(path, map_location="cpu")
    w = sd["token_embedding.weight"] if isinstance(sd, dict) and "token_embedding.weight" in sd else sd
    if w.shape[0] != n + 1:
        raise ValueError(f"{path}: token_embedding.weight has {w.shape[0]} rows, expected number_prefix_tokens + 1 = {n + 1}")
    return w[1:].float()


def generate_image(tr, tokens, noises, num_denoising_steps=30, prefix=None):
    """``generate_image`` of gen-images.py:112-175 (``generate_image_w_prefix_embedding`` :273-343 with ``prefix`` [n, D]; ``tokens`` then come
    from ``prefix_tokens``) on the trainer's no-grad rollout: tokens of ONE prompt (prompt ids/mask, uncond ids/mask), noises [N,4,h,w] ->
    images [N,3,H,W] in [-1,1] (fp16)."""
    enc = tr.encode_pair(tr.te, tokens, prefix=prefix)
    x, _, _ = tr.rollout(tr.unet, enc, noises.to(tr.device, torch.float32), num_denoising_steps)
    return tr.decode(x)


def main(args, cfgs=None):
    from .lib import WORKING_DTYPE
    if args.mixed_precision != WORKING_DTYPE:
        raise NotImplementedError(f"--mixed_precision {args.mixed_precision}: this process runs the {WORKING_DTYPE} library (fp16 and bf16 are "
                                  "built; select with FD_DTYPE or the flag on the command line; fp32 inference is not built)")
    if not torch.cuda.is_available():
        raise RuntimeError("finetune_fair_diffusion_amd.generate needs an MI355X (HIP device); there is no CPU path")
    from PIL import Image
    from .train import CLIPTokenizerAdapter, HashTokenizer
    device = torch.device("cuda", args.gpu_id)
    torch.cuda.set_device(device)
    cfgs = cfgs or (TINY if os.environ.get("FD_TINY") else SD15)
    targs = default_args(train_unet=bool(args.load_unet_lora_from), train_text_encoder=bool(args.load_text_encoder_lora_from),
                         rank=args.rank, guidance_scale=args.guidance_scale, pretrained_model_name_or_path=args.pretrained_model_name_or_path)
    state_dicts = None
    if not args.synthetic:
        from . import pretrained as P
        m = args.pretrained_model_name_or_path
        state_dicts = dict(unet=P.load_unet(m, cfgs["unet"]), vae=P.load_vae(m, cfgs["vae"]), clip=P.load_text_encoder(m, cfgs["clip"]))
    state_dicts = dict(state_dicts or {})
    if args.load_unet_lora_from:
        state_dicts["unet_lora"] = torch.load(args.load_unet_lora_from, map_location="cpu")
    if args.load_text_encoder_lora_from:
        state_dicts["te_lora"] = torch.load(args.load_text_encoder_lora_from, map_location="cpu")
    tr, _ = build_trainer(targs, device, cfgs, state_dicts=state_dicts, frozen_copies=False)
    tok_dir = os.path.join(args.pretrained_model_name_or_path, "tokenizer")
    tokenizer = CLIPTokenizerAdapter(tok_dir) if os.path.isdir(tok_dir) else HashTokenizer(cfgs["clip"].vocab_size)
    prefix = load_prefix_embedding(args.load_prefix_embedding_from, args.number_prefix_tokens) if args.load_prefix_embedding_from else None
    with open(args.prompts_path, "r") as f:
        test_prompts = json.load(f)["test_prompts"]
    lat = cfgs["unet"].sample_size
    written = []
    for i, prompt in enumerate(test_prompts):
        d = os.path.join(args.save_dir, f"prompt_{i}")
        os.makedirs(d, exist_ok=True)
        noises, paths = [], []
        for j in range(args.num_imgs_per_prompt):
            torch.manual_seed(args.random_seed + hash(prompt) + j)       # every (prompt, j) noise is drawn, used or not (:533-541)
            n = torch.randn([1, 4, lat, lat], dtype=torch.float32)
            path = os.path.join(d, f"img_{j}.jpg")
            if not os.path.exists(path):
                noises.append(n)
                paths.append(path)
        if not noises:
            continue
        noises = torch.cat(noises)
        tokens = tokenizer(prompt)
        if prefix is not None:
            tokens = prefix_tokens(tokens, args.number_prefix_tokens, cfgs["clip"].vocab_size)
        for b in range(math.ceil(len(paths) / args.batch_size)):
            nb = noises[b * args.batch_size:(b + 1) * args.batch_size].to(device)
            imgs = to_uint8_hwc(generate_image(tr, tokens, nb, args.num_denoising_steps, prefix=prefix))
            for img, path in zip(imgs, paths[b * args.batch_size:(b + 1) * args.batch_size]):
                Image.fromarray(img).save(path)
                written.append(path)
    return written


if __name__ == "__main__":
    main(parse_args())
