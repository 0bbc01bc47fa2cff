"""Trainer-state checkpoints and the exported LoRA files, in the reference's public format.

The reference writes rolling ``checkpoint_tmp-{step}`` / long-cadence ``checkpoint-{step}`` directories with
``accelerator.save_state`` (exp-1-debias-gender/1-main-debias.py:2050-2068, clean-up :120-137) and later converts one
of them with ``2-export-checkpoint.py:619-642`` into four files -- ``{text_encoder,unet}_lora{,_EMA}.pth`` -- each a
``torch.save``d ``dict[str, Tensor]`` (fp32, CPU) keyed by the diffusers LoRA names.  ``gen-images.py``
consumes exactly those four files.

Here a checkpoint directory *is* the exported format plus one extra file:

    checkpoint-{step}/unet_lora.pth, unet_lora_EMA.pth, text_encoder_lora.pth, text_encoder_lora_EMA.pth
    checkpoint-{step}/prefix_embedding.pth, prefix_embedding_EMA.pth      (exp-2 instead of the LoRA files)
    checkpoint-{step}/trainer_state.pth   Adam moments (flat fp32 per bank), optimiser / EMA / lr step counters, world size (rank 0 writes it)
    checkpoint-{step}/rng_rank{r}.pth     python / numpy / torch / OT-target RNG streams of rank r (every rank writes its own)

so the reference's export step becomes a file copy (``export_checkpoint``), and files exported by the reference load
back into the banks by key (``load_lora_files``).  Raw ``accelerator.save_state`` directories of the reference are read by
``accelerate_state.load_accelerate_state`` (``train --resume_from_checkpoint`` picks the reader by the directory's contents).
"""
import os
import random
import shutil

import numpy as np
import torch

BANK_FILES = {"unet": ("unet_lora.pth", "unet_lora_EMA.pth"), "text_encoder": ("text_encoder_lora.pth", "text_encoder_lora_EMA.pth"),
              # exp-2 (exp-2-debias-gender-token/2-export-checkpoint.py:566-575): the FairEmbeddings state dict, live and EMA
              "prefix_embedding": ("prefix_embedding.pth", "prefix_embedding_EMA.pth")}
# entries of a FairEmbeddings state dict that are not trained (buffers of the frozen text encoder): written for gen-images.py, ignored on load
PREFIX_FROZEN_KEYS = ("position_ids", "position_embedding.weight")


def clean_checkpoint(ckpts_save_dir, name, checkpoints_total_limit):
    """:120-137 -- before saving, keep at most ``limit - 1`` directories called ``{name}-{step}`` (oldest removed first).
    Deviation (harmless, deliberate): the reference matches ``d.startswith(name)``, so a call with name='checkpoint' would also count
    the rolling ``checkpoint_tmp-*`` directories (it only ever calls it with 'checkpoint_tmp', :2053); here the match is
    ``name + '-'`` followed by digits, so the two cadences can never delete each other's directories and ``*_exported`` copies stay."""
    ck = [d for d in os.listdir(ckpts_save_dir) if d.startswith(name + "-") and d.split("-")[1].isdigit()]   # "<ckpt>_exported" dirs stay
    ck = sorted(ck, key=lambda x: int(x.split("-")[1]))
    removed = []
    if len(ck) >= checkpoints_total_limit:
        removed = ck[0:len(ck) - checkpoints_total_limit + 1]
        for d in removed:
            shutil.rmtree(os.path.join(ckpts_save_dir, d))
    return removed


def trainer_banks(trainer):
    out = {}
    if getattr(trainer.args, "train_unet", False):
        out["unet"] = trainer.unet.lora_bank
    if getattr(trainer.args, "train_text_encoder", False):
        out["text_encoder"] = trainer.te.lora_bank
    if getattr(trainer, "prefix", None) is not None:
        out["prefix_embedding"] = trainer.prefix.bank
    return out


def save_lora_files(banks, path, prefix=None):
    """The four-file export (2-export-checkpoint.py:619-642): live and EMA weights by diffusers key, fp32 on CPU.  ``prefix``: the
    trainer's PrefixEmbedding (exp-2), whose files carry the full FairEmbeddings state dict."""
    os.makedirs(path, exist_ok=True)
    for which, bank in banks.items():
        live, ema = BANK_FILES[which]
        if which == "prefix_embedding" and prefix is not None:
            torch.save(prefix.state_dict(ema=False), os.path.join(path, live))
            torch.save(prefix.state_dict(ema=True), os.path.join(path, ema))
            continue
        torch.save(bank.state_dict(ema=False), os.path.join(path, live))
        torch.save(bank.state_dict(ema=True), os.path.join(path, ema))


def load_lora_files(banks, path, strict=True):
    """Loads ``*_lora.pth`` into the live weights and ``*_lora_EMA.pth`` (when present) into the EMA shadow."""
    for which, bank in banks.items():
        live, ema = BANK_FILES[which]
        sd = torch.load(os.path.join(path, live), map_location="cpu")
        missing = [n for n in bank.names if n not in sd]
        unexpected = [k for k in sd if k not in bank.offsets and not (which == "prefix_embedding" and k in PREFIX_FROZEN_KEYS)]
        if strict and (missing or unexpected):
            raise KeyError(f"{live}: missing {missing[:3]}... unexpected {unexpected[:3]}...")
        for n in bank.names:
            if n in sd:
                if tuple(sd[n].shape) != tuple(bank.shape(n)):
                    raise ValueError(f"{live}: {n} has shape {tuple(sd[n].shape)}, expected {tuple(bank.shape(n))}")
                bank.view(n).copy_(sd[n].to(bank.flat.device, torch.float32))
        bank.ema.copy_(bank.flat)
        p = os.path.join(path, ema)
        if os.path.exists(p):
            sde = torch.load(p, map_location="cpu")
            for n in bank.names:
                if n in sde:
                    bank.view(n, bank.ema).copy_(sde[n].to(bank.flat.device, torch.float32))


def _rng_state(trainer):
    return dict(python=random.getstate(), numpy=np.random.get_state(), torch=torch.get_rng_state(), targets=trainer.target_rng.get_state())


def save_state(trainer, save_path, global_step, extra=None):
    """What ``accelerator.save_state`` preserves for this loop: parameters, AdamW moments, scheduler position,
    the registered EMA models and the RNG streams (:1654-1659, :2058).  Called by EVERY rank: rank 0 writes the shared state,
    each rank writes its own ``rng_rank{r}.pth`` -- the per-rank streams (CPU noise :1746-1749 seeded ``seed + rank`` :693, OT draws
    1234 + rank) must stay distinct after a resume, as they do in the reference, where a rank that finds no ``random_states_{r}.pkl``
    keeps its own device-specific stream."""
    rank, world = getattr(trainer, "rank", 0), getattr(trainer, "world", 1)
    banks = trainer_banks(trainer)
    if rank == 0:
        save_lora_files(banks, save_path, prefix=getattr(trainer, "prefix", None))
        st = dict(global_step=int(global_step), opt_step=int(trainer.opt_step), lr_step=int(getattr(trainer, "lr_step", 0)),
                  ema_steps=[e.optimization_step for e in trainer.ema], bank_order=list(banks.keys()),
                  exp_avg={k: b.exp_avg.detach().cpu() for k, b in banks.items()},
                  exp_avg_sq={k: b.exp_avg_sq.detach().cpu() for k, b in banks.items()},
                  world_size=world, extra=extra or {})
        torch.save(st, os.path.join(save_path, "trainer_state.pth"))
    else:
        os.makedirs(save_path, exist_ok=True)
    torch.save(_rng_state(trainer), os.path.join(save_path, f"rng_rank{rank}.pth"))
    return save_path


def load_state(trainer, path, seed=None):
    """Inverse of save_state; returns the global step (the reference parses it from the directory name, :1708).  Every rank restores
    the shared state and ONLY ITS OWN RNG file; when the checkpoint was written by a different world size (or the file is absent)
    the rank keeps / re-derives its device-specific streams (``seed + rank``, OT 1234 + rank) instead of cloning rank 0's."""
    rank, world = getattr(trainer, "rank", 0), getattr(trainer, "world", 1)
    banks = trainer_banks(trainer)
    load_lora_files(banks, path)
    st = torch.load(os.path.join(path, "trainer_state.pth"), map_location="cpu", weights_only=False)
    if st["bank_order"] != list(banks.keys()):
        raise ValueError(f"checkpoint trains {st['bank_order']}, this run trains {list(banks.keys())}")
    for k, b in banks.items():
        b.exp_avg.copy_(st["exp_avg"][k].to(b.flat.device))
        b.exp_avg_sq.copy_(st["exp_avg_sq"][k].to(b.flat.device))
    trainer.opt_step = st["opt_step"]
    trainer.lr_step = st["lr_step"]
    for e, n in zip(trainer.ema, st["ema_steps"]):
        e.optimization_step = n
    rng_file = os.path.join(path, f"rng_rank{rank}.pth")
    rng = None
    if st.get("world_size", 1) == world and os.path.exists(rng_file):
        rng = torch.load(rng_file, map_location="cpu", weights_only=False)
    elif "rng" in st and world == 1:          # round-1 single-process checkpoints kept the streams inside trainer_state.pth
        rng = st["rng"]
    if rng is not None:
        random.setstate(rng["python"])
        np.random.set_state(rng["numpy"])
        torch.set_rng_state(rng["torch"])
        trainer.target_rng.set_state(rng["targets"])
    elif seed is not None:                     # different world size: fresh device-specific streams, offset by the step so they are new
        random.seed(seed + rank + 7919 * st["global_step"])
        np.random.seed((seed + rank + 7919 * st["global_step"]) % (2 ** 32))
        torch.manual_seed(seed + rank + 7919 * st["global_step"])
        trainer.target_rng.manual_seed(1234 + rank + 7919 * st["global_step"])
    if getattr(trainer.args, "train_unet", False):
        trainer.unet.refresh_lora()
    if getattr(trainer.args, "train_text_encoder", False):
        trainer.te.refresh_lora()
    return st["global_step"]


def export_checkpoint(resume_from_checkpoint):
    """2-export-checkpoint.py:598-642: ``<ckpt>`` -> ``<ckpt>_exported/`` holding only the LoRA files."""
    if not resume_from_checkpoint or not os.path.exists(resume_from_checkpoint):
        raise ValueError(f"{resume_from_checkpoint}' does not exist.")
    out = resume_from_checkpoint.rstrip("/") + "_exported"
    os.makedirs(out, exist_ok=True)
    done = []
    for files in BANK_FILES.values():
        for f in files:
            src = os.path.join(resume_from_checkpoint, f)
            if os.path.exists(src):
                shutil.copyfile(src, os.path.join(out, f))
                done.append(f)
    return out, done
