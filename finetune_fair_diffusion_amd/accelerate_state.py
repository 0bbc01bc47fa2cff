"""Reader for RAW ``accelerator.save_state`` directories of the reference (exp-1-debias-gender/1-main-debias.py:2050-2068; resumed with
``accelerator.load_state`` :1697-1724, exported by 2-export-checkpoint.py) -- so a run started with the reference can be continued here
without going through its export script.

What accelerate 0.27 writes for this training script, and what each piece becomes here:

    model.safetensors / pytorch_model.bin            1st prepared model   (``_1`` suffix: 2nd)
        text-encoder LoRA ``CustomModel`` (:856-872): keys ``params.<i>``, i in the order of the list ``_modify_text_encoder`` returns (attention q, k, v, out of all layers, then fc1, fc2 of all layers)  -> te.lora_bank
        U-Net ``AttnProcsLayers`` (:818): keys by attention-processor name (diffusers' state-dict hook)                   -> unet.lora_bank
      prepare order (:1653-1657): the text-encoder model first (when trained), then the U-Net layers
    optimizer.bin        torch AdamW ``state_dict``; parameter indices follow ``params_to_optimize`` (:889-895): U-Net parameters in
                         ``unet.attn_processors`` order (down, up, mid blocks; q, k, v, out; down, up), then the text-encoder ones   -> exp_avg / exp_avg_sq, step
    scheduler.bin        LambdaLR ``state_dict`` (``last_epoch``)                                                                   -> trainer.lr_step
    custom_checkpoint_<j>.pkl   the registered ``EMAModel``s (:1654, :1657), text encoder first: ``shadow_params`` list + ``optimization_step``  -> bank.ema, EMAState
    random_states_<rank>.pkl    python / numpy / torch CPU generator states of that rank                                           -> the process's generators

The global step is parsed from the directory name, as the reference does (:1708).  The OT-target generator (exp-3/4/5) is this build's own
and keeps its state.  Host-side glue: plain torch / safetensors file reading, no HIP.
"""
import os
import pickle
import random

import numpy as np
import torch

TE_ATTN, TE_MLP = ("q_proj", "k_proj", "v_proj", "out_proj"), ("fc1", "fc2")


def te_reference_param_order(num_layers):
    """Names of the text-encoder LoRA tensors in the order of the parameter LIST that ``LoraLoaderMixin._modify_text_encoder`` returns
    (diffusers 0.19.3; :831): the reference builds its CustomModel (``params.<i>``), the AdamW parameter indices and the EMA shadows by
    iterating over THAT list (:836-842 -- ``named_parameters()`` is only searched for each entry's name).  The list holds q, k, v, out_proj
    of every layer's attention first, then -- ``patch_mlp=True`` -- fc1, fc2 of every layer's MLP; a LoRALinearLayer yields ``down`` then ``up``."""
    out = []
    for i in range(num_layers):
        for tgt in TE_ATTN:
            p = f"text_model.encoder.layers.{i}.self_attn.{tgt}.lora_linear_layer."
            out += [p + "down.weight", p + "up.weight"]
    for i in range(num_layers):
        for tgt in TE_MLP:
            p = f"text_model.encoder.layers.{i}.mlp.{tgt}.lora_linear_layer."
            out += [p + "down.weight", p + "up.weight"]
    return out


def unet_reference_param_order(cfg):
    """Names of the U-Net LoRA tensors in ``AttnProcsLayers(unet.attn_processors).parameters()`` order: diffusers 0.19.3 registers
    ``down_blocks``, ``up_blocks``, then ``mid_block`` (the two ModuleLists are created before the mid block), a LoRAAttnProcessor its
    to_q_lora, to_k_lora, to_v_lora, to_out_lora, each ``down`` then ``up``."""
    n = cfg.layers_per_block
    procs = []
    for i, t in enumerate(cfg.down_block_types):
        if t.startswith("CrossAttn"):
            procs += [f"down_blocks.{i}.attentions.{j}.transformer_blocks.0.attn{a}.processor" for j in range(n) for a in (1, 2)]
    for i, t in enumerate(cfg.up_block_types):
        if t.startswith("CrossAttn"):
            procs += [f"up_blocks.{i}.attentions.{j}.transformer_blocks.0.attn{a}.processor" for j in range(n + 1) for a in (1, 2)]
    procs += [f"mid_block.attentions.0.transformer_blocks.0.attn{a}.processor" for a in (1, 2)]
    return [f"{p}.{w}_lora.{d}.weight" for p in procs for w in ("to_q", "to_k", "to_v", "to_out") for d in ("down", "up")]


def is_accelerate_state_dir(path):
    return os.path.isdir(path) and os.path.exists(os.path.join(path, "optimizer.bin")) and not os.path.exists(os.path.join(path, "trainer_state.pth"))


def _load_model_file(path, index):
    suf = "" if index == 0 else f"_{index}"
    st = os.path.join(path, f"model{suf}.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st)
    for name in (f"pytorch_model{suf}.bin", f"model{suf}.bin"):
        p = os.path.join(path, name)
        if os.path.exists(p):
            return torch.load(p, map_location="cpu", weights_only=False)
    raise FileNotFoundError(f"{path}: no model{suf}.safetensors / pytorch_model{suf}.bin")


def load_accelerate_state(trainer, path, restore_rng=True):
    """Restores LoRA parameters, AdamW moments and step, lr-scheduler position, EMA shadows and (this rank's) RNG streams from a raw
    ``accelerator.save_state`` directory; returns the global step.  Raises when the directory does not match what this run trains."""
    args = trainer.args
    train_te = bool(getattr(args, "train_text_encoder", False)) and trainer.te.lora_bank is not None
    train_unet = bool(getattr(args, "train_unet", False)) and trainer.unet.lora_bank is not None
    if getattr(trainer, "prefix", None) is not None:
        raise NotImplementedError("exp-2 (prefix embedding) accelerate states are not mapped; resume from this build's own checkpoints")
    if not (train_te or train_unet):
        raise ValueError("nothing trained: no LoRA bank to restore into")
    te_names = te_reference_param_order(trainer.te.config.num_hidden_layers) if train_te else []
    un_names = unet_reference_param_order(trainer.unet.config) if train_unet else []
    banks = {}
    if train_te:
        banks["text_encoder"] = (trainer.te.lora_bank, te_names)
        if set(te_names) != set(trainer.te.lora_bank.names):
            raise ValueError("text-encoder LoRA names of this build do not match the reference order table")
    if train_unet:
        banks["unet"] = (trainer.unet.lora_bank, un_names)
        if set(un_names) != set(trainer.unet.lora_bank.names):
            raise ValueError("U-Net LoRA names of this build do not match the reference order table")

    pending = []         # (bank, buffer, name, tensor): every shape is validated BEFORE the first copy, so a mismatch leaves the trainer untouched

    def put(bank, buf, name, t):
        if tuple(t.shape) != tuple(bank.shape(name)):
            raise ValueError(f"{path}: {name} has shape {tuple(t.shape)}, expected {tuple(bank.shape(name))} (different LoRA rank, or a "
                             f"parameter order this reader does not know)")
        pending.append((bank, buf, name, t))

    # ---- models, in prepare order: text encoder first (:1653), then the U-Net (:1656)
    mi = 0
    if train_te:
        sd = _load_model_file(path, mi); mi += 1
        if len(sd) != len(te_names) or any(f"params.{i}" not in sd for i in range(len(te_names))):
            raise ValueError(f"{path}: model file {mi - 1} is not the text-encoder CustomModel ({len(sd)} tensors, expected params.0..{len(te_names) - 1})")
        for i, n in enumerate(te_names):
            put(trainer.te.lora_bank, None, n, sd[f"params.{i}"])
    if train_unet:
        sd = _load_model_file(path, mi); mi += 1
        missing = [n for n in un_names if n not in sd]
        if missing or len(sd) != len(un_names):
            raise ValueError(f"{path}: model file {mi - 1} is not the U-Net AttnProcsLayers (missing {missing[:2]}, {len(sd)} tensors)")
        for n in un_names:
            put(trainer.unet.lora_bank, None, n, sd[n])

    # ---- optimizer: U-Net parameters first, then the text encoder's (:889-895)
    opt = torch.load(os.path.join(path, "optimizer.bin"), map_location="cpu", weights_only=False)
    order = [("unet", n) for n in un_names] + [("text_encoder", n) for n in te_names]
    pidx = [i for g in opt["param_groups"] for i in g["params"]]
    if len(pidx) != len(order):
        raise ValueError(f"{path}: optimizer holds {len(pidx)} parameters, this run trains {len(order)}")
    steps = set()
    for i, (which, n) in zip(pidx, order):
        st = opt["state"].get(i)
        bank = banks[which][0]
        if st is None:          # parameter never stepped
            put(bank, bank.exp_avg, n, torch.zeros(bank.shape(n))); put(bank, bank.exp_avg_sq, n, torch.zeros(bank.shape(n)))
            continue
        put(bank, bank.exp_avg, n, st["exp_avg"])
        put(bank, bank.exp_avg_sq, n, st["exp_avg_sq"])
        steps.add(int(st["step"]))
    if len(steps) > 1:
        raise ValueError(f"{path}: parameters carry different AdamW step counts {sorted(steps)}")
    opt_step = steps.pop() if steps else 0
    base = os.path.basename(os.path.normpath(path))
    global_step = int(base.split("-")[1]) if "-" in base and base.split("-")[1].isdigit() else 0

    # ---- lr scheduler.  The reference prepares it through accelerate (:1648) with split_batches=False: AcceleratedScheduler.step() advances the
    # inner LambdaLR ``num_processes`` times per optimiser step (which is why :1642-1643 scale warm-up and max steps by num_processes), so a
    # W-process run stores last_epoch = W * global_step.  This build's lr_lambda counts ONE unit per step with unscaled warm-up / max steps.
    lr_step = None
    sch = os.path.join(path, "scheduler.bin")
    if os.path.exists(sch):
        last_epoch = int(torch.load(sch, map_location="cpu", weights_only=False).get("last_epoch", 0))
        n_rng = len([f for f in os.listdir(path) if f.startswith("random_states_") and f.endswith(".pkl")])
        if global_step > 0 and last_epoch % global_step == 0:
            w_ref = last_epoch // global_step
        else:
            w_ref = max(n_rng, 1)
        if w_ref < 1 or last_epoch % w_ref != 0 or (n_rng and w_ref != n_rng and global_step > 0):
            raise ValueError(f"{path}: scheduler last_epoch {last_epoch} is not a multiple of the run's process count "
                             f"(global step {global_step}, {n_rng} random_states files)")
        lr_step = last_epoch // w_ref

    # ---- EMA models, registered text encoder first (:1654), then U-Net (:1657); trainer.banks is [unet, text_encoder] (step.py)
    ci = 0
    ema_steps = []
    for which in ("text_encoder", "unet"):
        if which not in banks:
            continue
        p = os.path.join(path, f"custom_checkpoint_{ci}.pkl"); ci += 1
        bank, names = banks[which]
        with open(p, "rb") as f:
            try:
                ema = torch.load(f, map_location="cpu", weights_only=False)
            except Exception:
                f.seek(0)
                ema = pickle.load(f)
        shadow = ema["shadow_params"]
        if len(shadow) != len(names):
            raise ValueError(f"{p}: {len(shadow)} shadow parameters, expected {len(names)}")
        for n, t in zip(names, shadow):
            put(bank, bank.ema, n, t)
        ema_steps.append((bank, int(ema.get("optimization_step", 0))))

    # ---- everything validated: commit
    for bank, buf, name, t in pending:
        bank.view(name, buf).copy_(t.to(bank.flat.device, torch.float32))
    trainer.opt_step = opt_step
    if lr_step is not None:
        trainer.lr_step = lr_step
    for bank, st in ema_steps:
        for b, e in zip(trainer.banks, trainer.ema):
            if b is bank:
                e.optimization_step = st

    # ---- RNG streams of this rank
    if restore_rng:
        rp = os.path.join(path, f"random_states_{getattr(trainer, 'rank', 0)}.pkl")
        if os.path.exists(rp):
            st = torch.load(rp, map_location="cpu", weights_only=False)
            if "random_state" in st:
                random.setstate(st["random_state"])
            if "numpy_random_seed" in st:
                np.random.set_state(st["numpy_random_seed"])
            if "torch_manual_seed" in st:
                torch.set_rng_state(st["torch_manual_seed"])
    if train_unet:
        trainer.unet.refresh_lora()
    if train_te:
        trainer.te.refresh_lora()
    return global_step
