"""Training driver: the loop of exp-1-debias-gender/1-main-debias.py ``main`` (:647-2070) around FairnessTrainer.

    python -m finetune_fair_diffusion_amd.train [--experiment exp-1|exp-2|exp-3|exp-4|exp-5] --config <yaml> [--synthetic]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m finetune_fair_diffusion_amd.train ...

Same flags, YAML overlay, seeding (``set_seed(seed, device_specific=True)`` :693, prompt order from
``random.seed(seed+1)`` :914-921, S drawn on rank 0 from range(19,24) and broadcast :1779-1781, per-rank CPU noise
:1746-1749), checkpoint cadence (:2050-2068) and resume (:1697-1725) as the reference; wandb/plots/evaluation grids are
replaced by one JSON line per step on rank 0.  One process per GPU; RCCL through torch.distributed ("nccl").
"""
import json
import math
import os
import random
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from . import checkpoint as ckpt
from .cli import parse_args
from .factory import SD15, TINY, build_trainer, synthetic_tokens

SYNTHETIC_PROMPTS = dict(prompt_templates_train=["a photo of the face of a {occupation}, a person", "a portrait of a {occupation}, a person"],
                         occupations_train_set=["doctor", "teacher", "engineer", "chef", "pilot", "nurse"])


class HashTokenizer:
    """Stand-in when no CLIP vocabulary is mounted (no network here): whitespace words -> stable ids in [320, 40000).
    Produces what the reference's tokenizer calls produce in shape (:1007, :1020-1026): prompt ``[BOS, w..., EOS]`` with an
    all-ones mask, and the empty prompt padded to the same length with EOS and mask ``[1,1,0,...]``."""

    def __init__(self, vocab=49408, max_len=77):
        self.vocab, self.max_len = vocab, max_len

    def __call__(self, prompt):
        import zlib
        bos, eos = self.vocab - 2, self.vocab - 1
        lo, hi = min(320, self.vocab // 4), min(40000, self.vocab - 2)
        words = prompt.replace(",", " ,").split()[: self.max_len - 2]
        ids = [bos] + [lo + zlib.crc32(w.lower().encode()) % (hi - lo) for w in words] + [eos]
        L = len(ids)
        return (torch.tensor(ids), torch.ones(L, dtype=torch.long), torch.tensor([bos] + [eos] * (L - 1)), torch.tensor([1, 1] + [0] * (L - 2)))


class CLIPTokenizerAdapter:
    """The reference's two tokenizer calls (:1003-1026) on a locally mounted ``<model>/tokenizer`` directory."""

    def __init__(self, path):
        from transformers import CLIPTokenizer
        self.tok = CLIPTokenizer.from_pretrained(path)

    def __call__(self, prompt):
        t = self.tok([prompt], return_tensors="pt", padding=True, truncation=True, max_length=self.tok.model_max_length)
        L = t.input_ids.shape[1]
        u = self.tok([""], return_tensors="pt", padding="max_length", truncation=True, max_length=L)
        return t.input_ids[0], t.attention_mask[0], u.input_ids[0], u.attention_mask[0]


def load_prompts(args):
    """Occupation prompts (:905-908); exp-5 mixes in three more files at 6x / 20x / 4x (exp-5 :934-947)."""
    if os.path.exists(args.prompt_occupation_path):
        with open(args.prompt_occupation_path, "r") as f:
            data = json.load(f)
    elif getattr(args, "synthetic", False):
        data = SYNTHETIC_PROMPTS
    else:
        raise FileNotFoundError(f"{args.prompt_occupation_path} (pass --synthetic to run without the reference's data.zip)")
    prompts = [p.format(occupation=o) for p in data["prompt_templates_train"] for o in data["occupations_train_set"]]
    extra = [("prompt_occupation_w_style_and_context_path", 6), ("prompt_personal_descroptor_path", 20), ("prompt_sports_path", 4)]
    if all(hasattr(args, k) for k, _ in extra):
        for k, rep in extra:
            path = getattr(args, k)
            if os.path.exists(path):
                with open(path, "r") as f:
                    prompts += json.load(f)["train_prompts"] * rep
            elif not getattr(args, "synthetic", False):
                raise FileNotFoundError(f"{path} (pass --synthetic to run without the reference's data.zip)")
    return prompts


def set_seed(seed, device_specific, rank):
    """accelerate.utils.set_seed as called at :693."""
    if device_specific:
        seed += rank
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def main(argv=None, experiment=None, cfgs=None, log=None):
    from_command_line = argv is None
    argv = list(argv if argv is not None else sys.argv[1:])
    if experiment is None:       # build addition: one driver for the reference's per-experiment scripts
        import argparse
        pre = argparse.ArgumentParser(add_help=False)
        pre.add_argument("--experiment", default="exp-1", choices=["exp-1", "exp-2", "exp-3", "exp-4", "exp-5"])
        ns, argv = pre.parse_known_args(argv)
        experiment = ns.experiment
    args = parse_args(argv, with_extras=True, experiment=experiment)
    args.experiment = experiment
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    from .lib import WORKING_DTYPE
    if args.mixed_precision not in ("fp16", "bf16"):
        raise NotImplementedError("--mixed_precision no (fp32 activations) is not built: the MFMA path computes in fp16 or bf16")
    if args.mixed_precision != WORKING_DTYPE:
        raise RuntimeError(f"--mixed_precision {args.mixed_precision} but this process was started with the {WORKING_DTYPE} library: the working "
                           "dtype is fixed at import: set FD_DTYPE=%s in the environment (python -m finetune_fair_diffusion_amd.train reads --mixed_precision "
                           "and the --config YAML by itself; the YAML value wins, as in the reference)" % args.mixed_precision)
    if from_command_line or world > 1:      # the command line (or a torchrun rank): an embedding caller that passes its own argv keeps its process's affinity and thread pool
        from .affinity import pin_rank
        pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))      # this rank's CPUs / intra-op threads, before the first HIP call
    if not torch.cuda.is_available():
        raise RuntimeError("finetune_fair_diffusion_amd.train needs an MI355X (HIP device); there is no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    set_seed(args.seed, True, rank)
    cfgs = cfgs or (TINY if os.environ.get("FD_TINY") else SD15)
    state_dicts = None
    if not args.synthetic:
        from .pretrained import load_pretrained
        state_dicts = load_pretrained(args, cfgs)
    face_provider = None
    if getattr(args, "face_provider", "synthetic") == "detector":       # the reference's insightface + face_recognition side-car (:936-945, :1192-1353)
        from .fairness import DetectorFaceProvider
        face_provider = DetectorFaceProvider.from_installed()
    elif getattr(args, "face_provider", "synthetic") != "synthetic":
        raise ValueError(f"--face_provider {args.face_provider}: 'synthetic' or 'detector'")
    trainer, models = build_trainer(args, device, cfgs, seed=args.seed, rank=rank, world_size=world, experiment=experiment, face_provider=face_provider,
                                    state_dicts=state_dicts, regularisers=(args.weight_loss_img != 0 or args.weight_loss_face != 0),
                                    lora_up_std=getattr(args, "lora_up_std", 0.0))
    tok_dir = os.path.join(args.pretrained_model_name_or_path, "tokenizer")
    tokenizer = CLIPTokenizerAdapter(tok_dir) if os.path.isdir(tok_dir) else HashTokenizer(cfgs["clip"].vocab_size)
    if not os.path.isdir(tok_dir) and not args.synthetic:
        raise FileNotFoundError(f"{tok_dir}: no CLIP vocabulary (pass --synthetic for the hash tokenizer)")

    prompts_train = load_prompts(args)
    steps_per_epoch = len(prompts_train)
    num_epochs = math.ceil(args.max_train_steps / steps_per_epoch)
    random.seed(args.seed + 1)   # identical prompt order on every rank (:914-921)
    order = []
    for _ in range(num_epochs):
        idxs = list(range(steps_per_epoch))
        random.shuffle(idxs)
        order.append(idxs)

    ckpts_dir = os.path.join(args.output_dir, "checkpoints")
    if rank == 0:
        os.makedirs(ckpts_dir, exist_ok=True)
    global_step, first_epoch, resume_step = 0, 0, 0
    if args.resume_from_checkpoint:
        if not os.path.exists(args.resume_from_checkpoint):
            if rank == 0:
                print(f"Checkpoint '{args.resume_from_checkpoint}' does not exist. Starting a new training run.")
            args.resume_from_checkpoint = None
        else:
            from . import accelerate_state
            if accelerate_state.is_accelerate_state_dir(args.resume_from_checkpoint):     # a directory written by the reference's accelerator.save_state (:2058)
                global_step = accelerate_state.load_accelerate_state(trainer, args.resume_from_checkpoint)
            else:
                global_step = ckpt.load_state(trainer, args.resume_from_checkpoint, seed=args.seed)
            first_epoch, resume_step = global_step // steps_per_epoch, global_step % steps_per_epoch
    lat = cfgs["unet"].sample_size
    B = args.train_images_per_prompt_GPU
    def draw(data_idx):
        """The per-step host inputs in the reference's order (:1746-1749, :1779): prompt, CPU noise, number of denoising steps."""
        prompt = prompts_train[data_idx]
        noises = torch.randn([B, 4, lat, lat], dtype=torch.float32)          # CPU generator, differs by rank (:1746-1749)
        S = [args.num_denoising_steps or random.choices(range(19, 24), k=1)[0]]
        if world > 1:
            dist.broadcast_object_list(S, src=0)
        return prompt, noises, S[0]

    def peek(data_idx):
        """What ``draw`` WILL return for the next step, without consuming the generators: the draw is made on saved generator states that are
        restored afterwards, so the run's random streams -- and the RNG state a checkpoint stores -- are exactly those of a loop without look-ahead."""
        st = (random.getstate(), np.random.get_state(), torch.get_rng_state())
        try:
            return draw(data_idx)
        finally:
            random.setstate(st[0]); np.random.set_state(st[1]); torch.set_rng_state(st[2])

    def r2_tokens(toks):
        # exp-2 (:1954): the original side sees the plain prompt, its empty prompt encoded without a padding mask
        return (toks[0], toks[1], toks[2], torch.ones_like(toks[3])) if trainer.prefix is not None else toks

    plan = [(epoch, step, data_idx) for epoch in range(first_epoch, num_epochs) for step, data_idx in enumerate(order[epoch])
            if not (args.resume_from_checkpoint and epoch == first_epoch and step < resume_step)]
    for i, (epoch, step, data_idx) in enumerate(plan):
        if global_step >= args.max_train_steps:
            break
        prompt, noises, S0 = draw(data_idx)
        S = [S0]
        t0 = time.time()
        toks = tokenizer(prompt)
        # the next step's inputs, so that its frozen-model rollout can start underneath this step's tail (step.py, r2_prefetch_steps)
        nxt = None
        if i + 1 < len(plan) and global_step + 1 < args.max_train_steps and os.environ.get("FD_NO_R2_PREFETCH") is None:
            p_n, noises_n, S_n = peek(plan[i + 1][2])
            nxt = dict(tokens_ori=r2_tokens(tokenizer(p_n)), noises=noises_n, S=S_n)
        if trainer.prefix is not None:
            # exp-2 (:1846, :1895, :1954, :2001): the finetuned side sees "".join(prefix_tokens) + prompt with the pipeline's negative
            # prompt (no padding mask); the original side sees the plain prompt, its empty prompt encoded without a mask as well
            from .generate import prefix_tokens
            out = trainer.train_step(prefix_tokens(toks, trainer.prefix.n, cfgs["clip"].vocab_size), noises, S[0],
                                     tokens_ori=r2_tokens(toks), next_step=nxt)
        else:
            out = trainer.train_step(toks, noises, S[0], next_step=nxt)
        global_step += 1
        if rank == 0:
            lf = out["loss_fair"]
            rec = dict(step=global_step, prompt=prompt, S=S[0], noise_checksum=float(noises.double().sum()), lr=trainer.last_lr, grad_is_finite=out["grad_is_finite"],
                       loss_fair=float(lf[lf != -1].mean()) if bool((lf != -1).any()) else None,
                       loss_CLIP=float(out["loss_CLIP"].mean()) if "loss_CLIP" in out else None,
                       loss_DINO=float(out["loss_DINO"].mean()) if "loss_DINO" in out else None,
                       loss_face=(float(out["loss_face"][out["loss_face"] != -1].mean()) if (out["loss_face"] != -1).any() else None) if "loss_face" in out else None,
                       p_class1_mean=float(out["probs"][:, 1][out["probs"][:, 1] != -1].mean()) if bool((out["probs"] != -1).any()) else None,
                       seconds=round(time.time() - t0, 3))
            (log or print)(json.dumps(rec))
        # checkpoints (:2050-2068): rank 0 cleans up and writes the shared state, every rank adds its own RNG streams
        if global_step % args.checkpointing_steps == 0:
            if rank == 0 and args.checkpoints_total_limit is not None:
                ckpt.clean_checkpoint(ckpts_dir, "checkpoint_tmp", args.checkpoints_total_limit)
            if world > 1:
                dist.barrier()
            ckpt.save_state(trainer, os.path.join(ckpts_dir, f"checkpoint_tmp-{global_step}"), global_step)
        if global_step % args.checkpointing_steps_long == 0:
            ckpt.save_state(trainer, os.path.join(ckpts_dir, f"checkpoint-{global_step}"), global_step)
    if world > 1:
        dist.barrier()
    return trainer, global_step


if __name__ == "__main__":
    main()
