"""Loading real weights from a locally mounted Stable-Diffusion-v1.5 directory in the diffusers layout

    <model>/unet/diffusion_pytorch_model.{safetensors,bin}     <model>/vae/diffusion_pytorch_model.{safetensors,bin}
    <model>/text_encoder/{model.safetensors,pytorch_model.bin}  <model>/tokenizer/   (see train.CLIPTokenizerAdapter)

(what ``from_pretrained(..., subfolder=...)`` reads at exp-1-debias-gender/1-main-debias.py:734-749) plus the
attribute classifier's torchvision state dict (:929-935).  There is no hub download here: the directory must exist.
The modules of this package consume tensors by their diffusers / transformers-4.30 / torchvision keys, so loading is a
key filter + shape check; two historical renamings are handled:

* VAE mid-block attention saved with the pre-0.18 names ``query/key/value/proj_attn`` -> ``to_q/to_k/to_v/to_out.0``
* CLIP text weights saved by newer transformers without the ``text_model.`` prefix
"""
import math
import os

import torch

from . import weights as W

_VAE_LEGACY = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def _read(path_noext_candidates):
    for p in path_noext_candidates:
        if os.path.exists(p):
            if p.endswith(".safetensors"):
                from safetensors.torch import load_file
                return load_file(p, device="cpu")
            return torch.load(p, map_location="cpu")
    raise FileNotFoundError(" | ".join(path_noext_candidates))


def _select(sd, shapes, what, rename=None):
    out, missing = {}, []
    src = {}
    for k, v in sd.items():
        k2 = rename(k) if rename else k
        src[k2] = v
    for k, shp in shapes.items():
        if k not in src:
            missing.append(k)
            continue
        v = src[k]
        if v.numel() != math.prod(shp):
            raise ValueError(f"{what}: {k} has shape {tuple(v.shape)}, expected {tuple(shp)}")
        out[k] = v.reshape(shp).float()
    if missing:
        raise KeyError(f"{what}: {len(missing)} tensors missing, e.g. {missing[:4]}")
    return out


def _vae_rename(k):
    parts = k.split(".")
    if "attentions" in parts and len(parts) >= 2 and parts[-2] in _VAE_LEGACY:
        parts[-2:-1] = _VAE_LEGACY[parts[-2]].split(".")
    return ".".join(parts)


def _clip_rename(k):
    return k if k.startswith("text_model.") else "text_model." + k


def load_unet(model_dir, cfg):
    d = os.path.join(model_dir, "unet")
    sd = _read([os.path.join(d, "diffusion_pytorch_model.safetensors"), os.path.join(d, "diffusion_pytorch_model.bin")])
    return _select(sd, W.unet_param_shapes(cfg), "unet")


def load_vae(model_dir, cfg):
    d = os.path.join(model_dir, "vae")
    sd = _read([os.path.join(d, "diffusion_pytorch_model.safetensors"), os.path.join(d, "diffusion_pytorch_model.bin")])
    return _select(sd, W.vae_param_shapes(cfg), "vae", _vae_rename)      # decoder + post_quant_conv only; encoder keys ignored


def load_text_encoder(model_dir, cfg):
    d = os.path.join(model_dir, "text_encoder")
    sd = _read([os.path.join(d, "model.safetensors"), os.path.join(d, "pytorch_model.bin")])
    return _select(sd, W.clip_param_shapes(cfg), "text_encoder", _clip_rename)


def load_classifier(path, num_classes):
    sd = torch.load(path, map_location="cpu")
    sd = sd.get("state_dict", sd)
    sd = {k[len("model."):] if k.startswith("model.") else k: v for k, v in sd.items()}
    return _select(sd, W.mobilenet_param_shapes(num_classes), "classifier")


def load_pretrained(args, cfgs):
    """-> dict for factory.build_trainer(state_dicts=...)."""
    from .fairness import EXPERIMENT_ATTRS
    m = args.pretrained_model_name_or_path
    if not os.path.isdir(m):
        raise FileNotFoundError(f"{m}: not a local diffusers directory (no network here; pass --synthetic for synthetic weights)")
    experiment = getattr(args, "experiment", "exp-1")
    out = dict(unet=load_unet(m, cfgs["unet"]), vae=load_vae(m, cfgs["vae"]), clip=load_text_encoder(m, cfgs["clip"]))
    out["clf"] = load_classifier(args.classifier_weight_path, EXPERIMENT_ATTRS[experiment][0])
    return out
