"""Loading real weights from a locally mounted Stable-Diffusion-v1.5 directory in the diffusers layout

    <model>/unet/diffusion_pytorch_model.{safetensors,bin}     <model>/vae/diffusion_pytorch_model.{safetensors,bin}
    <model>/text_encoder/{model.safetensors,pytorch_model.bin}  <model>/tokenizer/   (see train.CLIPTokenizerAdapter)

(what ``from_pretrained(..., subfolder=...)`` reads at exp-1-debias-gender/1-main-debias.py:734-749) plus the
attribute classifier's torchvision state dict (:929-935) and the regularisers' assets: the opensphere SFNet-20 backbone and
``face_feats.pkl`` (:968-994, paths from the reference's own flags) and CLIP ViT-H/14 / DINOv2 ViT-B/14 (:948-964; hub downloads in the
reference, local paths from ``FD_CLIP_VISION_DIR`` / ``FD_DINO_WEIGHTS`` here).  There is no hub download here: the directory must exist.
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


def load_face_net(path, in_size=112):
    """opensphere ``backbone_*.pth`` (:979-985): saved from ``nn.DataParallel``, keys carry a ``module.`` prefix."""
    sd = torch.load(path, map_location="cpu")
    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
    return _select(sd, W.sfnet20_param_shapes(in_size=in_size), "face_net (sfnet20)")


def load_face_db(path):
    """FaceFeatsModel.__init__ (:83-91): ``face_feats.pkl`` = (face_feats [n,512], genders, logits); rows L2-normalised."""
    import pickle
    with open(path, "rb") as f:
        obj = pickle.load(f)
    feats = obj[0] if isinstance(obj, (tuple, list)) else obj
    feats = torch.as_tensor(feats).float()
    if feats.dim() != 2 or feats.shape[1] != 512:
        raise ValueError(f"{path}: face features have shape {tuple(feats.shape)}, expected [n, 512]")
    return torch.nn.functional.normalize(feats, dim=-1)


def load_clip_vision(model_dir, cfg):
    """A local ``CLIPVisionModelWithProjection`` directory (the hub snapshot of laion/CLIP-ViT-H-14-laion2B-s32B-b79K, :948-951);
    full CLIP checkpoints carry the text tower too, which the key filter drops."""
    sd = _read([os.path.join(model_dir, "model.safetensors"), os.path.join(model_dir, "pytorch_model.bin"),
                os.path.join(model_dir, "open_clip_pytorch_model.bin")])
    return _select(sd, W.vit_param_shapes(cfg), "clip_vision")


def load_dino(path, cfg):
    """``dinov2_vitb14_pretrain.pth`` as torch.hub stores it (:958-961); ``mask_token`` is unused at inference and dropped."""
    sd = torch.load(path, map_location="cpu")
    return _select(sd, W.vit_param_shapes(cfg), "dino")


def load_regularisers(args, cfgs):
    """Weights of the loss regularisers for factory.build_trainer(state_dicts=...).  The reference fetches CLIP ViT-H/14 and DINOv2
    from the hub caches; there is no network here, so their locations come from ``FD_CLIP_VISION_DIR`` / ``FD_DINO_WEIGHTS``."""
    out = {}
    if getattr(args, "weight_loss_img", 0) != 0:
        cdir, dpath = os.environ.get("FD_CLIP_VISION_DIR"), os.environ.get("FD_DINO_WEIGHTS")
        if not cdir or not dpath:
            raise FileNotFoundError("weight_loss_img != 0 needs FD_CLIP_VISION_DIR (CLIP ViT-H/14 directory) and FD_DINO_WEIGHTS "
                                    "(dinov2_vitb14_pretrain.pth); there is no hub access here (pass --synthetic for synthetic weights)")
        out["clip_vision"] = load_clip_vision(cdir, cfgs["clip_vision"])
        out["dino"] = load_dino(dpath, cfgs["dino"])
    if getattr(args, "weight_loss_face", 0) != 0:
        out["face_net"] = load_face_net(args.opensphere_model_path, in_size=args.size_aligned_face)
        out["face_db"] = load_face_db(args.face_feats_path)
    return out


def load_pretrained(args, cfgs):
    """-> dict for factory.build_trainer(state_dicts=...)."""
    from .fairness import EXPERIMENT_ATTRS
    m = args.pretrained_model_name_or_path
    if not os.path.isdir(m):
        raise FileNotFoundError(f"{m}: not a local diffusers directory (no network here; pass --synthetic for synthetic weights)")
    experiment = getattr(args, "experiment", "exp-1")
    out = dict(unet=load_unet(m, cfgs["unet"]), vae=load_vae(m, cfgs["vae"]), clip=load_text_encoder(m, cfgs["clip"]))
    out["clf"] = load_classifier(args.classifier_weight_path, EXPERIMENT_ATTRS[experiment][0])
    out.update(load_regularisers(args, cfgs))
    return out
