"""The fairness-finetuning training step on one MI355X rank -- the hot path of
exp-1-debias-gender/1-main-debias.py:1746-2029 re-designed for this hardware:

* R1 / R2: no-grad CFG rollouts (``generate_image_no_gradient`` :998-1061) of the finetuned and the
  frozen original models: S fused U-Net forwards on the 2N CFG batch, one fused CFG+DPM-Solver++
  update kernel per step, VAE decode.
* R3: ``generate_image_w_gradient`` (:1063-1136) without an autograd graph: dL/dx_final is obtained once
  (classifier + VAE backward), and because the U-Net input is detached at every step the gradient of eps_i is
  the scalar ``grad_coef_i * c_i`` times dL/dx_final (scheduler.chain_coefs) -- every timestep is back-propagated
  independently.  The forward rollout keeps each timestep's activations in HBM when they fit (7.8 GB per
  timestep at batch 8; all 20 fit in 288 GB) and otherwise only its input latent [N,4,64,64], in which case the
  timestep is recomputed with recording right before its backward (the reference's gradient checkpointing, :748).
* all micro-batches of the reference (``train_GPU_batch_size`` chunks, :1889) run as ONE batch with
  per-image weights 1/n_j (identical gradient, see fairness.microbatch_weights).
* gradient sync: one RCCL all-reduce of the flat fp32 LoRA-gradient buffer + fused scale/finite
  check + one AdamW+EMA launch (replaces the 400 per-tensor collectives/launches of :1998-2029).

The loss is ``loss_fair + weight_loss_img * dynamic_weights * (loss_CLIP + loss_DINO)`` (:1904-1932) when the two image
encoders are attached (``clip_model`` / ``dino_model``, vit.py), plus ``weight_loss_face * loss_face`` (:1917-1932) when the
face-feature network and its feature database are attached (``face_net`` sfnet.py, ``face_db``); all input gradients join the
classifier's at the image and go through the VAE once.  A non-zero ``weight_loss_face`` without a face network is refused
rather than silently dropped.
"""
import contextlib
import math
import os
import threading
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import ops
from .fairness import (EXPERIMENT_ATTRS, EXPERIMENT_REG_FLAGS, SyntheticFaceProvider, face_grad_factors, face_grad_factors_multi,
                       fair_loss_and_grad, gen_dynamic_weights, gen_dynamic_weights_multi,
                       generate_dynamic_targets, mc_transport_plan, microbatch_weights, targets_from_plan)
from . import fairness_dev as FD
from . import unet as unet_mod
from .layers import F16, F32
from .lr_schedule import lr_lambda
from .vit import feature_loss_and_grad

# The two halves of a CFG batch share everything up to the first cross-attention (unet.forward_step(pair=True)); FD_NO_CFG_PAIR=1
# evaluates the duplicated batch instead (A/B measurement only; eps is bit-identical either way).
_CFG_PAIR = os.environ.get("FD_NO_CFG_PAIR") is None
_ROCTX = os.environ.get("FD_ROCTX") is not None


def _ctx_droppable_bytes(ctx):
    """Bytes of one timestep's context that lean recording does not keep (unet.TransformerBlock: the pre-gate FF projection, n1, n2)."""
    if isinstance(ctx, dict):
        n = sum(t.numel() * t.element_size() for k in ("proj", "n1", "n2") for t in (ctx.get(k),) if isinstance(t, torch.Tensor))
        return n + sum(_ctx_droppable_bytes(v) for v in ctx.values() if isinstance(v, (list, tuple)))
    if isinstance(ctx, (list, tuple)):
        return sum(_ctx_droppable_bytes(c) for c in ctx)
    return 0


def _ctx_bytes(obj, seen=None):
    """Bytes held by the tensors reachable from a recorded-forward context (unique storages)."""
    seen = set() if seen is None else seen
    if torch.is_tensor(obj):
        key = obj.untyped_storage().data_ptr()
        if key in seen:
            return 0
        seen.add(key)
        return obj.untyped_storage().nbytes()
    if isinstance(obj, dict):
        return sum(_ctx_bytes(v, seen) for v in obj.values())
    if isinstance(obj, (list, tuple)):
        return sum(_ctx_bytes(v, seen) for v in obj)
    return 0


def _record_stream(obj, stream):
    """Tell the caching allocator that the tensors of a recorded-forward context are also used on ``stream`` (they were allocated on
    the main stream): their memory is not handed out again before that stream's pending kernels have read them."""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)


def _all_tensors(obj, out=None):
    """Every tensor reachable from a recorded-forward context (the race detector's ``bwd_keep_alive`` holds them past the backward)."""
    out = [] if out is None else out
    if torch.is_tensor(obj):
        out.append(obj)
    elif isinstance(obj, dict):
        for v in obj.values():
            _all_tensors(v, out)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _all_tensors(v, out)
    return out


def _pow2_scale(amax, target):
    if not math.isfinite(amax) or amax <= 0:
        return 1.0
    return float(2.0 ** round(math.log2(target / amax)))


def _pow2_scale_dev(amax, target):
    """``_pow2_scale`` evaluated ON THE DEVICE (``amax``: device tensor of any shape -> same-shape fp32 power of two, 1 where amax is 0 or not
    finite): the loss scales of the VAE / ViT / SFNet / U-Net backwards no longer bounce through the host, so the tail of the step has no
    read-back between the two logits copies and the end of the step (FD_HOST_SCALES=1 restores the four host syncs for A/B)."""
    a = amax.float()
    ok = torch.isfinite(a) & (a > 0)
    s = torch.exp2(torch.round(torch.log2(target / torch.where(ok, a, torch.ones_like(a)))))
    return torch.where(ok, s, torch.ones_like(s))


_HOST_SCALES = os.environ.get("FD_HOST_SCALES") is not None
# FD_HOST_TAIL=1 (A/B): exp-1's targets / loss assembly on the HOST as in rounds 1-3 (two logits read-backs in the tail); default: on the device
_HOST_TAIL = os.environ.get("FD_HOST_TAIL") is not None
_R2_SIDE_OLD = os.environ.get("FD_R2_SIDE_HOST_ORDER") is not None
# FD_ATOMIC_DKDV=1 (A/B): shared cross-attention dK / dV accumulated with fp32 atomics across samples, timesteps and streams, as in rounds 1-3
_ATOMIC_DKDV = os.environ.get("FD_ATOMIC_DKDV") is not None
_NO_PINNED_H2D = os.environ.get("FD_NO_PINNED_H2D") is not None
_FINE_MARKS = os.environ.get("FD_FINE_MARKS") is not None
_MAIN_PRIORITY = int(os.environ.get("FD_MAIN_PRIORITY", "0"))
# FD_TAIL_REORDER=1 (measurement): enqueue the recorded CLIP / DINO forward of R1's images and R2's feature encoders AHEAD of the logits' read-backs.
# Measured slower (1409-1424 vs 1378-1402 ms, profiles/r03_step_ab_tail_reorder_rejected.txt): the work it moves in front of the first read-back
# delays everything behind it by its full device time, while in the old order it hides behind the host-paced loss phase.
_NO_TAIL_REORDER = os.environ.get("FD_TAIL_REORDER") is None


def _h2d(t, dev):
    """Small host tensor -> device without a blocking copy: staged through pinned memory (torch's caching host allocator) and copied
    non-blocking; ``t.to(dev)`` from pageable memory is a synchronous hipMemcpy behind everything queued on the launch stream.  The loss
    phase has nine such copies between its read-backs (loss phase 62-78 -> 55-58 ms; the whole step within noise,
    profiles/r03_step_ab_pinned_h2d.txt)."""
    if _NO_PINNED_H2D:          # measurement switch (FD_NO_PINNED_H2D=1): the old pageable copies
        return t.to(dev).contiguous()
    return t.contiguous().pin_memory().to(dev, non_blocking=True)


class EMAState:
    """diffusers ``EMAModel`` decay schedule (no warm-up flag): decay_n = min(decay, (1+n)/(10+n)), first step copies."""

    def __init__(self, decay):
        self.decay, self.optimization_step = decay, 0

    def next_one_minus_decay(self):
        self.optimization_step += 1
        step = max(0, self.optimization_step - 1)
        d = 0.0 if step <= 0 else min((1 + step) / (10 + step), self.decay)
        return 1.0 - d


class FairnessTrainer:
    def __init__(self, args, text_encoder, unet, vae, classifier, scheduler, eval_text_encoder=None, eval_unet=None,
                 face_provider=None, experiment="exp-1", rank=0, world_size=1, device=None, clip_model=None, dino_model=None,
                 face_net=None, face_db=None, prefix_embedding=None):
        self.args = args
        self.prefix = prefix_embedding      # exp-2: trainable prompt-prefix vectors (prefix.PrefixEmbedding); every network stays frozen
        self.clip, self.dino = clip_model, dino_model
        self.use_img_loss = clip_model is not None and dino_model is not None and getattr(args, "weight_loss_img", 0) != 0
        if (clip_model is None) != (dino_model is None):
            raise ValueError("the image-semantics term needs both encoders (CLIP and DINOv2) or neither")
        self.face_net = face_net
        self.face_db = None if face_db is None else F.normalize(face_db.to(unet.device if device is None else device, F32), dim=-1)  # :88
        self.face_db16 = None if face_db is None else self.face_db.to(F16).contiguous()
        self.use_face_loss = face_net is not None and face_db is not None and getattr(args, "weight_loss_face", 0) != 0
        if (face_net is None) != (face_db is None):
            raise ValueError("the face-realism term needs both the face-feature network and its feature database, or neither")
        if self.use_img_loss and not self.use_face_loss and getattr(args, "weight_loss_face", 0) != 0:
            raise ValueError("weight_loss_face != 0 but no face-feature network / database is attached "
                             "(build_trainer(..., regularisers=True) attaches them; or pass --weight_loss_face 0)")
        f1, f2, conf = EXPERIMENT_REG_FLAGS[experiment]
        if self.use_img_loss or self.use_face_loss:
            self.factors1, self.factors2 = [getattr(args, k) for k in f1], [getattr(args, k) for k in f2]
            self.face_conf = getattr(args, conf)
        self.te, self.unet, self.vae, self.clf, self.sch = text_encoder, unet, vae, classifier, scheduler
        self.eval_te = eval_text_encoder if eval_text_encoder is not None else text_encoder
        self.eval_unet = eval_unet if eval_unet is not None else unet
        self.faces = face_provider or SyntheticFaceProvider()
        # which classifier columns carry which attribute (exp-1: CelebA attr 20 of 40x2 logits `:1370`; exp-3/5: 2+4; exp-4: 2+4+2)
        self.experiment = experiment
        _, self.attrs, self.class_cdfs, self.age_asym = EXPERIMENT_ATTRS[experiment]
        # exp-1 (one binary attribute): probabilities, dynamic targets, loss, dynamic weights and hook factors stay ON THE DEVICE (fairness_dev.py);
        # what the caller is told about the step (probabilities, targets, per-image losses) comes back in ONE read-back together with the
        # finite flag.  The multi-attribute experiments keep the host path (their OT solve has its own worker thread / device solver).
        self.device_tail = len(self.attrs) == 1 and self.attrs[0][2] == 2 and not _HOST_TAIL
        self._binom = {}
        self._pending_readback = None
        self.target_rng = torch.Generator().manual_seed(1234 + rank)
        self.rank, self.world = rank, world_size
        # collectives run whenever there is more than one rank -- or, with FD_FORCE_COLLECTIVES=1 and an initialised process group, on a
        # single rank too (the only way to drive the RCCL code path on a one-GPU box: RCCL refuses two ranks on one device)
        self.collectives = world_size > 1 or (os.environ.get("FD_FORCE_COLLECTIVES") is not None and dist.is_available() and dist.is_initialized())
        self._hbm_reserve = int(os.environ.get("FD_HBM_RESERVE_MB", "2048")) << 20      # since round 6 also without collectives: the runtime itself needs room (HSA_STATUS_ERROR_OUT_OF_RESOURCES at 0 free)
        self.device = device or unet.device
        self.banks = []
        if getattr(args, "train_unet", False):
            self.banks.append(unet.lora_bank)
        if getattr(args, "train_text_encoder", False):
            self.banks.append(text_encoder.lora_bank)
        if self.prefix is not None:
            if self.banks:
                raise ValueError("exp-2 trains the prefix embedding only (1-main-debias.py:946): no LoRA banks next to it")
            self.banks.append(self.prefix.bank)
        self.ema = [EMAState(args.EMA_decay) for _ in self.banks]
        self.opt_step = 0
        self.lr_step = 0      # lr_scheduler.step() count: advances every step, also when the update is skipped (:2023)
        self.flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.clf_gscale = 1024.0
        # R3 keeps per-timestep activations in HBM when they fit (MI355X: 288 GB) instead of recomputing every
        # timestep; set keep_activations=False for the pure recompute schedule of the reference (:748).
        self.keep_activations = True
        self.activation_mem_fraction = 0.85
        self.last_ctx_bytes = self.last_ctx_budget = 0
        self._full_ctx_bytes = 0              # bytes of one timestep's FULL context (measured by the first recording rollout)
        self._auto_lean = False              # the automatic mode chose lean recording once: it stays (see rollout_steps)
        # None: automatic (lean recording when S full contexts do not fit, from the second step on); True / False force it (FD_LEAN_ACTIVATIONS=1 / 0)
        self.lean_activations = {"1": True, "0": False}.get(os.environ.get("FD_LEAN_ACTIVATIONS", ""), None)
        # R1 and the forward half of R3 evaluate the same function on the same inputs (same prompt, noise, S and LoRA
        # weights; only the grad bookkeeping differs in the reference).  With deterministic, batch-invariant kernels the
        # two are bit-identical (asserted in tests), so R3 can consume R1's recorded rollout/decode/classifier forward.
        # ON by default whenever R1 runs as one chunk (val_GPU_batch_size >= B) and a separate frozen U-Net serves R2; FD_NO_SHARE=1
        # executes R1 and R3 separately, exactly like the reference (A/B measurement; same images, same gradient to rounding).
        self.share_r1_r3 = os.environ.get("FD_NO_SHARE") is None
        # per-phase wall-clock of the last step (HIP events on the launch stream; read with phase_ms()); None = off
        self.timers = None
        self._marks, self._host_marks = [], []
        # exp-3/4/5: run the Monte-Carlo OT solves on a worker thread underneath R2 (False = inline, like the reference)
        self.overlap_targets = os.environ.get("FD_NO_OT_OVERLAP") is None
        # R1 and R2 rollouts enqueued in lockstep on two HIP streams (FD_NO_CONCURRENT_R2=1: one after the other, as the reference does)
        self.concurrent_r2 = os.environ.get("FD_NO_CONCURRENT_R2") is None
        # backward of odd timesteps on the side stream (FD_NO_CONCURRENT_BWD=1: all on one stream)
        self.concurrent_bwd = os.environ.get("FD_NO_CONCURRENT_BWD") is None
        self.bwd_streams = int(os.environ.get("FD_BWD_STREAMS", "3"))     # measured: 2 -> 1647, 3 -> 1589, 4 -> 1633 ms per step (run-to-run noise ~2 %)
        self._side = None
        # race-detector knobs (tests / scratch/diag_hazard.py; never set by the product): ``bwd_virtual`` deals the timesteps to the same per-stream
        # gradient buffers but enqueues all of them on the launch stream -- same fp32 summation order as the concurrent schedule, so every
        # buffer must come out BIT-identical; ``bwd_keep_alive`` holds every consumed activation context until the backward has been joined;
        # ``debug_partials`` (a list) receives clones of the per-stream buffers before they are summed.
        self.bwd_virtual = False
        self.bwd_keep_alive = False
        self.debug_partials = None
        # R2 of the NEXT step does not depend on this step's update (frozen original models, its own noise and prompt): when the caller hands
        # over the next step's inputs (``train_step(..., next_step=...)``) its first denoising steps are enqueued on the R2 stream as soon as
        # this step's R2 has finished, i.e. underneath the VAE decode / classifier / loss / VAE backward tail, whose launches leave most of the
        # chip idle (host syncs, small kernels).  Same kernels on the same inputs: results are bit-identical with and without it.
        # same-box A/Bs.  Round 3: 0 -> 1442-1448 ms, 6 -> 1431-1436, 8 -> 1429, 10 -> 1440.  End of round 5 (medians of 6-step runs,
        # profiles/r05_step_ab_schedule_knobs_final_tree.txt): 0 -> 1317-1326, 2 -> 1309-1310, 4 -> 1307-1310, 5 -> 1309-1315, 6 -> 1309-1313, 7 -> 1315-1320, 8 -> 1316-1320,
        # 10 -> 1323-1325, 12 -> 1334-1338: the flat region moved down as the tail got shorter; 5 sits in its middle
        self.r2_prefetch_steps = int(os.environ.get("FD_R2_PREFETCH_STEPS", "5"))
        # ... and ``r2_prefetch_late`` more of them are enqueued behind the U-Net backward's last timestep: they run while the backward streams
        # drain unevenly, through the optimiser step and under the first (host-paced) forward of the next step
        self.r2_prefetch_late = int(os.environ.get("FD_R2_PREFETCH_LATE", "0"))
        # the frozen model's forward (R2: a third of the step's U-Net passes) replayed as a hipGraph: bit-identical to the eager forward, ~2900 C-ABI
        # calls per forward become one launch -- the enqueue thread keeps a larger lead over the device in the rollout phase, where its lead is smallest
        # (fewer of the +50..95 ms steps, median -5..-15 ms in 20-step runs: profiles/r04_step_jitter_r2_graph.txt).  FD_R2_GRAPH=0 turns it off.
        self.r2_graph = os.environ.get("FD_R2_GRAPH", "1") != "0"
        self._r2_pre = None
        self._sch_r2 = None
        self.last_r2_prefetched = 0
        self.last_ot_ms = (0.0, 0.0)
        # the Monte-Carlo transport solves of exp-3/4/5 run on the GPU (csrc/assign.hip); FD_OT_HOST=1: the host solver (measurement switch)
        self.ot_on_device = os.environ.get("FD_OT_HOST") is None
        self._ot_stream = torch.cuda.Stream(device=self.device)
        self._tgt = None

    # ------------------------------------------------------------------ per-phase timing (R1 / R2 / R3-fwd / R3-bwd / sync)
    def _mark(self, name):
        """Phase boundary: a HIP event on the launch stream (no host sync) and, with FD_ROCTX=1, a roctx range for rocprofv3 --marker-trace."""
        if self.timers is None:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self._marks.append((name, ev))
        self._host_marks.append((name, time.perf_counter()))
        if _ROCTX:
            if len(self._marks) > 1:
                torch.cuda.nvtx.range_pop()
            if name != "end":
                torch.cuda.nvtx.range_push(name)

    def _fine(self, name):
        """Sub-phase mark, only with FD_FINE_MARKS=1 (diagnosis of the host-bound tail; the bench line then carries the extra keys)."""
        if _FINE_MARKS:
            self._mark(name)

    def host_phase_ms(self):
        """{phase: ms} the HOST spent between the phase marks of the last step: launch-enqueue time (plus whatever host syncs the phase
        contains).  A phase whose host time equals its device time is launch-bound."""
        out = {}
        for (n0, t0), (_, t1) in zip(self._host_marks[:-1], self._host_marks[1:]):
            out[n0] = out.get(n0, 0.0) + (t1 - t0) * 1e3
        return out

    def phase_ms(self):
        """{phase: ms} of the last train_step (time from each mark to the next one), after a device sync."""
        torch.cuda.synchronize()
        out = {}
        for (n0, e0), (_, e1) in zip(self._marks[:-1], self._marks[1:]):
            out[n0] = out.get(n0, 0.0) + e0.elapsed_time(e1)
        return out

    def _side_stream(self, k=1):
        if self._side is None:
            self._side = {}
        if k not in self._side:
            # FD_R2_PRIORITY (measurement): HIP priority of the frozen-model rollout's stream (-1 high, 0 normal, 1 low where the runtime has it)
            prio = int(os.environ.get("FD_R2_PRIORITY", "0")) if k == "r2" else 0
            self._side[k] = torch.cuda.Stream(device=self.device, priority=prio) if prio else torch.cuda.Stream(device=self.device)
        return self._side[k]

    # ------------------------------------------------------------------ pieces
    def encode_pair(self, te, tokens, record=False, prefix=None):
        """tokens = (prompt_ids [L], prompt_mask [L], uncond_ids [L], uncond_mask [L]) -> enc [2,L,D] fp16, uncond first (:1035).
        ``prefix`` [n, D]: learned prefix-token embeddings for positions 1..n of the PROMPT sequence (exp-2 consumer)."""
        pid, pm, uid, um = tokens
        ids = torch.stack([uid, pid]).to(self.device)
        mask = torch.stack([um, pm]).to(self.device)
        return te.forward(ids, mask, record=record, prefix=None if prefix is None else (1, prefix))[0]

    def rollout_steps(self, unet, enc, noises, S, res, keep_inputs=False, record_prompt=False, keep_activations=False, sch=None):
        """CFG denoising rollout (:1038-1056) as a generator: yields after the launches of each denoising step so that two rollouts (R1
        on the current stream, R2 on the side stream) can be enqueued in lockstep; fills ``res`` = dict(lat, inputs, ctxs).
        With ``keep_activations`` the per-step backward contexts are kept for as many timesteps as fit in HBM
        (288 GB holds the whole 20-step chain at batch 8); the remaining steps are recomputed in the backward."""
        sch = self.sch if sch is None else sch       # a prefetched R2 rollout of the NEXT step brings its own scheduler object (S may differ)
        graphed = None
        if (self.r2_graph and unet is self.eval_unet and unet is not self.unet and unet.lora_bank is None and _CFG_PAIR
                and not keep_inputs and not keep_activations and not record_prompt and torch.cuda.current_stream() != torch.cuda.default_stream()):
            graphed = getattr(unet, "graphed", None) or unet_mod.GraphedForward(unet)
        sch.set_timesteps(S)
        unet.prepare_timesteps(sch.timesteps)
        unet.prepare_prompt(enc, record=record_prompt)
        if keep_activations and self.lean_activations is not False:
            # Lean recording (round 6; the reference's own memory lever is recompute, :748): when S timesteps of full contexts do not fit -- known from the previous
            # step's measured context size -- the transformer blocks keep half as much per token (unet.lean_record) and the backward recomputes three LayerNorms
            # and the FF1 projection per block instead of recomputing WHOLE timesteps: at S = 50, B = 8 that keeps ~45 of 50 timesteps instead of 28.
            full = self._full_ctx_bytes
            need_lean = self.lean_activations is True or (full > 0 and S * full > self.activation_mem_fraction * self._usable_free_bytes())
            if self.lean_activations is None and (need_lean or self._auto_lean):
                need_lean = self._auto_lean = True   # sticky in the automatic mode: a run whose S varies around the limit must not rebuild the allocator's pools every step
                                             # (a switch costs a device sync + ~10 s of pool refill; lean recording at a smaller S costs ~2 ms per timestep)
            if need_lean != unet.lean_record:
                # the caching allocator's pools hold blocks in the OTHER mode's sizes (28 timesteps x 16 pre-gate projections of 335 MB ...): they would sit
                # unused while the new mode's blocks come fresh from the driver until nothing is left for the runtime itself (HSA_STATUS_ERROR_OUT_OF_RESOURCES,
                # seen at S = 50 on the first lean step).  Hand them back once, at the switch (a device sync + a few ms; never in steady state).
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                self._snap_calls = 0
            unet.lean_record = need_lean
        lat = noises.clone()
        state, inputs, ctxs = {}, [], {}
        res.update(lat=lat, inputs=inputs, ctxs=ctxs)
        gs = self.args.guidance_scale
        budget = 0
        for i in range(S):
            if keep_inputs:
                inputs.append(lat.clone())
            rec = keep_activations and (i == 0 or budget > 0)
            if graphed is not None:        # the frozen model's non-recording forward as ONE hipGraph launch (unet.GraphedForward)
                eps = graphed(lat, i, _CFG_PAIR)
                sch.cfg_step(i, eps, gs, lat, state)
                yield i
                continue
            x = ops.to_f16(lat)
            eps = unet.forward_step(x if _CFG_PAIR else x.repeat(2, 1, 1, 1), i, record=rec, pair=_CFG_PAIR)   # cat([latents] * 2) (:1043)
            if rec:
                ctxs[i] = unet._ctx
                unet._ctx = None
                if i == 0:
                    per = _ctx_bytes(ctxs[0])
                    usable = self.activation_mem_fraction * self._usable_free_bytes()
                    if not unet.lean_record:
                        self._full_ctx_bytes = per
                        if self.lean_activations is None and S * per > usable:
                            # the very first recording rollout of a run (context size unknown until now) and S full contexts do not fit: timestep 0 stays as
                            # recorded, the remaining ones are recorded lean (the backward takes either form per timestep) -- no full-mode step that fills the HBM
                            # to the last GiB before the switch (S = 50 soak of round 6: 0.66 GiB of driver-free memory at step 0)
                            unet.lean_record = self._auto_lean = True
                            per -= _ctx_droppable_bytes(ctxs[0])
                    budget = int(usable / max(per, 1))
                    self.last_ctx_bytes, self.last_ctx_budget = per, budget
                else:
                    budget -= 1
            sch.cfg_step(i, eps, gs, lat, state)
            yield i

    def _usable_free_bytes(self):
        """HBM the recording rollout (on the CURRENT stream) can still obtain without the allocator having to synchronise: what the driver
        reports free plus the cached, unallocated blocks of the caching allocator -- minus the part of that cache which sits in the pools
        of the side streams (R2 rollout, backward streams): a block freed on another stream is only handed to this one after an OOM-retry
        ``free_cached_blocks`` + device sync, i.e. a stall in the middle of the step.  The side-stream share is estimated
        from the allocator's per-stream segment snapshot."""
        free, _ = torch.cuda.mem_get_info()
        cached = torch.cuda.memory_reserved() - torch.cuda.memory_allocated()
        cur = torch.cuda.current_stream().cuda_stream
        n = getattr(self, "_snap_calls", 0)
        self._snap_calls = n + 1
        reserved = torch.cuda.memory_reserved()
        grown = abs(reserved - getattr(self, "_snap_reserved", -1)) > (1 << 30)      # the pools moved by > 1 GiB since the last walk (a new S / batch,
                                                                                      # fragmentation in a long run): the side-stream share is stale
        if n < 2 or n % 256 == 0 or grown:
            # the pools settle within two steps; the snapshot walks every block of the allocator (tens of ms of host time at 175 GB) right at
            # the start of a step, where the device queue is empty: refreshed rarely -- and whenever ``memory_reserved`` has moved
            self._snap_reserved = reserved
            self._snap_walks = getattr(self, "_snap_walks", 0) + 1
            other = 0
            try:
                for seg in torch.cuda.memory_snapshot():
                    if seg.get("stream", cur) != cur:
                        other += sum(b["size"] for b in seg.get("blocks", ()) if b.get("state") == "inactive")
            except Exception:
                other = cached // 2
            self._other_stream_cache = other
        # a multi-rank run shares the device with RCCL: its communicator is resident before the first step (factory.build_trainer broadcasts the LoRA init),
        # so ``free`` already excludes it; what arrives later -- channel buffers of collectives first used inside the step, the probability gather's
        # staging -- is covered by a fixed reserve (FD_HBM_RESERVE_MB; measured resident set of the single-rank RCCL soak: profiles/r05_soak_s50_collectives.txt)
        return max(free - self._hbm_reserve, 0) + max(cached - self._other_stream_cache, 0)

    def rollout(self, unet, enc, noises, S, keep_inputs=False, record_prompt=False, keep_activations=False):
        """noises [N,4,h,w] fp32 on device.  Returns (x_final, [x_i], {i: ctx})."""
        res = {}
        for _ in self.rollout_steps(unet, enc, noises, S, res, keep_inputs, record_prompt, keep_activations):
            pass
        return res["lat"], res["inputs"], res["ctxs"]

    def decode(self, lat, record=False):
        return self.vae.decode_images(lat * (1.0 / self.vae.config.scaling_factor), record=record)

    def classify_begin(self, images, record=False):
        """First half of ``classify``: face boxes from the provider, crop + resize, classifier forward -- everything ENQUEUED, nothing read
        back, so that the caller can enqueue more device work that does not need the logits before it blocks in ``classify_end``."""
        N = images.shape[0]
        ind, boxes = self.faces(images)
        sel = ind.nonzero().view(-1)
        logits_dev = None
        if len(sel):
            chips = ops.crop_resize(images[_h2d(sel, images.device)].contiguous() if len(sel) != N else images, _h2d(boxes[sel], self.device), -1.0,
                                    self.args.size_face)
            logits_dev = self.clf.forward(chips, record=record).float()
        return dict(N=N, ind=ind, boxes=boxes, sel=sel, logits_dev=logits_dev)

    def classify_end(self, h):
        """Second half: the read-back of the logits (a host sync with the launch stream) and the per-attribute host tensors."""
        N, ind, boxes, sel, logits_dev = h["N"], h["ind"], h["boxes"], h["sel"], h["logits_dev"]
        per = []
        logits = logits_dev.cpu() if logits_dev is not None else None
        if self.collectives:
            # exchange point 1 stays on the device: [N, sum k] probabilities (-1 rows = no face) for ONE all-gather of all attributes
            pd = torch.full((N, sum(k for _, _, k in self.attrs)), -1.0, dtype=F32, device=self.device)
            if logits_dev is not None:
                pd[_h2d(sel, self.device)] = torch.cat([torch.softmax(logits_dev[:, c0:c0 + k], dim=-1) for _, c0, k in self.attrs], dim=1)
            self._probs_dev = pd
        for name, c0, k in self.attrs:
            probs = torch.full((N, k), -1.0)
            preds = torch.full((N,), -1, dtype=torch.long)
            la_full = torch.full((N, k), -1.0)
            if logits is not None:
                la = logits[:, c0:c0 + k]
                la_full[sel] = la
                p = torch.softmax(la, dim=-1)
                probs[sel] = p
                preds[sel] = p.max(dim=-1).indices
            per.append(dict(name=name, preds=preds, probs=probs, logits=la_full))
        return ind, boxes, per

    def _r2_side_forwards(self, images_ori, B, consumer):
        """Classifier forward and regulariser features of the frozen side's images, enqueued on the CURRENT (R2) stream; the tensors are handed to
        ``consumer`` (the launch stream waits on the event recorded behind them)."""
        r = dict(h_o=self.classify_begin(images_ori))
        if self.use_img_loss:                                                    # :1860-1862
            e_co, e_do = self.image_features(self.resize_small(images_ori)[0])
            r.update(clip_ori=F.normalize(e_co, dim=-1), dino_ori=F.normalize(e_do, dim=-1))
        if self.use_face_loss:                                                   # :1870
            from .sfnet import face_features
            ch_o, idx_o, _ = self.aligned_faces(images_ori, r["h_o"]["ind"])
            face_ori = torch.zeros((B, 512), dtype=F32, device=self.device)
            if len(idx_o):
                face_ori[idx_o.long()] = F.normalize(face_features(self.face_net, ch_o)[0], dim=-1)
            r["face_ori"] = face_ori
        for t in (r["h_o"]["logits_dev"], r.get("clip_ori"), r.get("dino_ori"), r.get("face_ori")):
            if t is not None:
                t.record_stream(consumer)
        return r

    def classify_dev(self, h):
        """``classify_end`` without the read-back (exp-1, ``device_tail``): the per-attribute tensors stay on the device."""
        N, ind, boxes, sel, logits_dev = h["N"], h["ind"], h["boxes"], h["sel"], h["logits_dev"]
        name, c0, k = self.attrs[0]
        if logits_dev is None:
            probs = torch.full((N, k), -1.0, dtype=F32, device=self.device)
            preds = torch.full((N,), -1, dtype=torch.long, device=self.device)
            lg = probs.clone()
        else:
            probs, preds, lg = FD.probs_preds(logits_dev, _h2d(sel, self.device), N, c0, k)
        return ind, boxes, [dict(name=name, preds=preds, probs=probs, logits=lg, ind_dev=_h2d(ind, self.device))]

    def _binom_tables(self, n):
        if n not in self._binom:
            self._binom[n] = tuple(t.to(self.device) for t in FD.binomial_tables(n))
        return self._binom[n]

    def classify(self, images, record=False):
        """get_face + get_face_gender[_race[_age]] (:1794-1795; exp-3 :1387-1457; exp-4 :1378-1475).
        Returns indicators, boxes and per attribute (preds [N], probs [N,k] (-1 filled), logits [N,k])."""
        return self.classify_end(self.classify_begin(images, record=record))

    def resize_small(self, images):
        """``transforms.Resize(img_size_small)`` (:1860, :1905): bilinear, no antialias == the crop kernel on the full-image box."""
        N, _, H, W = images.shape
        box = torch.tensor([[0, 0, W, H]] * N, dtype=torch.int32, device=self.device)
        return ops.crop_resize(images, box, -1.0, self.args.img_size_small), box

    def image_features(self, small, record=False):
        """get_clip_feat / get_dino_feat (:1139-1175) raw embeddings (fp32) of both encoders."""
        return self.clip.forward(small, record=record), self.dino.forward(small, record=record)

    def aligned_faces(self, images, ind):
        """aligned_face_chips of get_face (:1337-1338, image_pipeline :292-312) for the images with a face.
        Returns (chips [n,3,112,112] fp16, src_index [n] int32, A [n,6] fp32) -- the last two drive the backward scatter."""
        from .fairness import alignment_sampling_matrix
        import numpy as np
        N, _, H, W = images.shape
        crop = self.args.size_aligned_face
        lms = self.faces.landmarks(images)
        sel = ind.nonzero().view(-1).tolist()
        A = np.stack([alignment_sampling_matrix(lms[i].numpy(), H, W, crop) for i in sel]) if sel else np.zeros((0, 6))
        A = _h2d(torch.tensor(A, dtype=F32), self.device)
        idx = _h2d(torch.tensor(sel, dtype=torch.int32), self.device)
        return ops.warp_affine(images, idx, A, crop), idx, A

    def nearest_face_feats(self, query):
        """FaceFeatsModel.semantic_search (:98-117): database row with the largest dot product.  fp16 MFMA scores shortlist 8
        candidates per query; the winner is chosen among them in fp32."""
        M = self.face_db.shape[0]
        q16 = torch.zeros(((query.shape[0] + 7) // 8 * 8, query.shape[1]), dtype=F16, device=self.device)
        q16[:query.shape[0]] = query.to(F16)
        scores = ops.gemm(self.face_db16, q16, out_dtype=F32)[:, :query.shape[0]].t()         # [n, M]
        cand = scores.topk(min(8, M), dim=-1).indices                                           # [n, 8]
        exact = (self.face_db[cand] * query[:, None, :]).sum(dim=-1)
        best = cand.gather(1, exact.argmax(dim=-1, keepdim=True))[:, 0]
        return self.face_db[best]

    def start_dynamic_targets(self, per, B):
        """Global dynamic targets for this rank's B images from the gathered probabilities of all ranks (:1831-1837; exp-3 :2016-2025),
        first half: ONE all-gather for all attributes (device tensors in, one host copy out), then
          * exp-1: the binomial-rank targets, inline (microseconds);
          * exp-3/4/5: the 100 Monte-Carlo transport solves of this rank start on a WORKER THREAD -- they only need the gathered
            probabilities, and nothing needs the targets before R3's loss, so they run underneath the R2 rollout that the main thread
            keeps enqueueing (the reference solves them serially between R1 and R2 on every rank, exp-3 `:1488-1536`)."""
        if getattr(self, "device_tail", False) and per[0]["probs"].is_cuda:
            pd = per[0]["probs"]
            if self.collectives:
                gl = [torch.empty_like(pd) for _ in range(self.world)]
                dist.all_gather(gl, pd.contiguous())
                pd = torch.cat(gl)
            t_all, u_all = FD.dynamic_targets(pd, self._binom_tables(pd.shape[0]), threshold=self.args.uncertainty_threshold)
            self._tgt = dict(B=B, dev=[(t_all[B * self.rank:B * (self.rank + 1)], u_all[B * self.rank:B * (self.rank + 1)])])
            return
        if not self.collectives:
            gathered = [a["probs"] for a in per]
        else:
            allp = self.gather_probs(self._probs_dev)                  # [world*B, sum k] on the host
            gathered, c = [], 0
            for _, _, k in self.attrs:
                gathered.append(allp[:, c:c + k].contiguous())
                c += k
        self._tgt = dict(B=B, single=len(per) == 1)
        if len(per) == 1:
            self._tgt["res"] = [generate_dynamic_targets(gathered[0], w_uncertainty=True)]
            return

        st = self._tgt

        def work():
            t0 = time.perf_counter()
            try:
                if self.ot_on_device:       # fd_ot_assign_sum on a side stream of this thread; the summed plan stays in HBM for the all-reduce
                    with torch.cuda.stream(self._ot_stream):
                        idx, tp, sizes = mc_transport_plan(gathered, self.class_cdfs, 100, self.target_rng, self.age_asym, device=self.device)
                        if tp is not None and not self.collectives:
                            tp = tp.cpu()           # read back on THIS stream: a copy on the launch stream would wait for the queued rollout
                        st["plan"] = (idx, tp, sizes)
                    self._ot_stream.synchronize()
                else:
                    st["plan"] = mc_transport_plan(gathered, self.class_cdfs, 100, self.target_rng, self.age_asym)
            except BaseException as e:      # re-raised on the main thread by finish_dynamic_targets (a dead worker would otherwise surface
                st["error"] = e             # as KeyError('plan') here and as a collective timeout on the other ranks)
            st["solve_ms"] = 1e3 * (time.perf_counter() - t0)
        if self.overlap_targets:
            th = threading.Thread(target=work, daemon=True)
            th.start()
            self._tgt["thread"] = th
        else:
            work()

    def finish_dynamic_targets(self):
        """Second half: join the solver, ONE all-reduce of the summed plans (exchange point c11), marginals, threshold, this rank's slice.
        ``last_ot_ms`` = (host solve time, time the main thread actually waited here)."""
        st, args, B = self._tgt, self.args, self._tgt["B"]
        if "dev" in st:
            self._tgt = None
            return st["dev"]
        t0 = time.perf_counter()
        if not st["single"]:
            if "thread" in st:
                st["thread"].join()
            failed = torch.tensor([1.0 if "error" in st else 0.0])
            if self.collectives:            # every rank learns of a failed solve BEFORE the plan all-reduce, so all of them raise together
                f = failed.to(self.device)
                dist.all_reduce(f, op=dist.ReduceOp.MAX)
                failed = f.cpu()
            if float(failed) != 0:
                self._tgt = None
                raise RuntimeError("Monte-Carlo transport solve for the dynamic targets failed" +
                                   (" on this rank" if "error" in st else " on another rank")) from st.get("error")
            idx, tp, sizes = st["plan"]
            if tp is not None and tp.is_cuda:
                tp.record_stream(torch.cuda.current_stream())
            if tp is not None and self.collectives:
                t = tp.to(self.device)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                tp = t
            st["res"] = targets_from_plan(idx, tp, sizes)
            self.last_ot_ms = (st.get("solve_ms", 0.0), 1e3 * (time.perf_counter() - t0))
        out = []
        for t, u in st["res"]:
            t = t.clone()
            t[u > args.uncertainty_threshold] = -1
            out.append((t[B * self.rank:B * (self.rank + 1)], u[B * self.rank:B * (self.rank + 1)]))
        self._tgt = None
        return out

    def dynamic_targets(self, per, B):
        self.start_dynamic_targets(per, B)
        return self.finish_dynamic_targets()

    # ------------------------------------------------------------------ the two exchange points of the step
    def gather_probs(self, probs):
        """(1) all-gather of the per-rank class probabilities [B,k] so every rank derives identical global targets
        (:1805-1837; the reference also gathers images/boxes/preds for plotting only)."""
        if not getattr(self, "collectives", self.world > 1):
            return probs
        pg = probs.to(self.device).contiguous()
        gl = [torch.empty_like(pg) for _ in range(self.world)]
        dist.all_gather(gl, pg)
        return torch.cat(gl).cpu()          # concatenated on the device, ONE copy to the host (it was one copy per rank)

    def allreduce_grads(self):
        """(2) ONE all-reduce(SUM) per flat fp32 LoRA-gradient buffer (RCCL over xGMI on the GPU box)."""
        if getattr(self, "collectives", self.world > 1):
            for bank in self.banks:
                dist.all_reduce(bank.grad, op=dist.ReduceOp.SUM)

    # ------------------------------------------------------------------ the step
    def train_step(self, tokens, noises, S, tokens_ori=None, next_step=None):
        """One training step (``_train_step`` below).  With FD_MAIN_PRIORITY=-1 (measurement) the whole step is launched from a HIGH-priority stream
        instead of the caller's current one, so that the critical path's many small tail kernels are dispatched ahead of the frozen-model
        rollout's (normal-priority) kernels they share the chip with."""
        if _MAIN_PRIORITY == 0:
            return self._train_step(tokens, noises, S, tokens_ori, next_step)
        if getattr(self, "_main_stream", None) is None:
            self._main_stream = torch.cuda.Stream(device=self.device, priority=_MAIN_PRIORITY)
        caller = torch.cuda.current_stream()
        self._main_stream.wait_stream(caller)
        with torch.cuda.stream(self._main_stream):
            out = self._train_step(tokens, noises, S, tokens_ori, next_step)
        caller.wait_stream(self._main_stream)
        return out

    def _train_step(self, tokens, noises, S, tokens_ori=None, next_step=None):
        """``tokens``: the prompt the finetuned side sees (exp-2: ``prompt_debiaser(prompt)``, generate.prefix_tokens); ``tokens_ori``: the
        prompt of the frozen original side R2 when it differs (exp-2 :1954: the plain prompt, no prefix embedding).
        ``next_step``: optional dict(tokens_ori=, noises=, S=) -- the inputs the NEXT call will receive (noises as a host tensor): the first
        ``r2_prefetch_steps`` denoising steps of its R2 rollout are then enqueued underneath this step's tail."""
        args = self.args
        tokens_ori = tokens if tokens_ori is None else tokens_ori
        dev = self.device
        B = noises.shape[0]
        pre, self._r2_pre = self._r2_pre, None
        if pre is not None and not (pre["S"] == S and not noises.is_cuda and pre["noises_host"].shape == noises.shape and
                                    torch.equal(pre["noises_host"], noises.to(F32)) and
                                    all(torch.equal(a, b) for a, b in zip(pre["tokens_ori"], tokens_ori))):
            pre = None              # the caller changed its mind: the prefetched steps are dropped (their kernels were harmless)
        if pre is not None:
            torch.cuda.current_stream().wait_event(pre["ev_noise"])
            noises = pre["noises_dev"]
            noises.record_stream(torch.cuda.current_stream())
        else:
            noises = noises.to(dev, F32)
        out = {}
        for bank in self.banks:
            bank.grad.zero_()
        vb = args.val_GPU_batch_size
        self._marks, self._host_marks = [], []
        self._mark("R1_rollout")
        # ---- R1: images from the model being finetuned (:1786-1795)
        train_te = getattr(args, "train_text_encoder", False) and self.te.lora_bank is not None
        train_unet = getattr(args, "train_unet", False) and self.unet.lora_bank is not None
        train_prefix = self.prefix is not None
        pv = self.prefix.vectors() if train_prefix else None
        rec_te = train_te or train_prefix
        share = self.share_r1_r3 and vb >= B and self.eval_unet is not self.unet and (train_unet or train_te or train_prefix)
        shared = None
        # R1 (finetuned model) and R2 (frozen original, :1844-1858) are independent until the loss: with ``concurrent_r2`` their denoising
        # steps are enqueued in lockstep on two HIP streams, so the many launches that cannot fill 256 CUs on their own (16x16 / 8x8 levels,
        # tail waves, latency-bound short-K GEMMs) overlap with the other rollout's work.  Same kernels, same results.
        conc = self.concurrent_r2 and vb >= B and self.eval_unet is not self.unet
        cur = torch.cuda.current_stream()
        side = self._side_stream("r2") if conc else None      # its own stream: the backward's side streams must not queue behind a prefetch
        r2 = {}
        self.last_r2_prefetched = 0
        if conc and pre is not None:
            g2, r2 = pre["gen"], pre["res"]                       # already ``pre["k"]`` denoising steps ahead
            self.last_r2_prefetched = pre["k"]
        elif conc:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                enc_ori = self.encode_pair(self.eval_te, tokens_ori)
                g2 = self.rollout_steps(self.eval_unet, enc_ori, noises, S, r2)
        if share:
            enc = self.encode_pair(self.te, tokens, record=rec_te, prefix=pv)
            r1 = {}
            g1 = self.rollout_steps(self.unet, enc, noises, S, r1, keep_inputs=True, record_prompt=True, keep_activations=self.keep_activations)
            for _ in g1:
                if conc:
                    with torch.cuda.stream(side):
                        next(g2, None)
            x_final, inputs, ctxs = r1["lat"], r1["inputs"], r1["ctxs"]
            self._mark("R1_vae")
            images = self.decode(x_final, record=True)
            shared = (enc, inputs, ctxs)
        else:
            enc = self.encode_pair(self.te, tokens, prefix=pv)
            lats = []
            for j in range(0, B, vb):
                r1 = {}
                for _ in self.rollout_steps(self.unet, enc, noises[j:j + vb], S, r1):
                    if conc:
                        with torch.cuda.stream(side):
                            next(g2, None)
                lats.append(r1["lat"])
            self._mark("R1_vae")
            images = torch.cat([self.decode(x) for x in lats])
        if conc:
            with torch.cuda.stream(side):
                for _ in g2:           # (nothing left when both rollouts have S steps)
                    pass
                images_ori = self.decode(r2["lat"])
                # the frozen side's classifier and regulariser forwards (:1860-1870) depend on images_ori only: enqueued HERE, on the R2 stream, they run
                # beside R1's last denoising steps instead of on the launch stream between R1's classifier and the loss (FD_R2_SIDE_HOST_ORDER=1: as before)
                r2side = None
                if not _R2_SIDE_OLD:
                    r2side = self._r2_side_forwards(images_ori, B, cur)
            ev_r2 = torch.cuda.Event()
            ev_r2.record(side)         # what the main stream waits for below: this step's R2, not a prefetch queued behind it
            can_prefetch = (next_step is not None and self.r2_prefetch_steps > 0 and not next_step["noises"].is_cuda and
                            next_step["noises"].shape[0] <= vb and not (train_te and self.eval_te is self.te))
            if can_prefetch:
                import copy
                if self._sch_r2 is None:
                    self._sch_r2 = copy.deepcopy(self.sch)
                nt = next_step.get("tokens_ori")
                k = min(self.r2_prefetch_steps, int(next_step["S"]))
                with torch.cuda.stream(side):
                    nh = next_step["noises"].to(F32)
                    nd = nh.to(dev, non_blocking=False)
                    ev_noise = torch.cuda.Event()
                    ev_noise.record(side)
                    r2n = {}
                    g2n = self.rollout_steps(self.eval_unet, self.encode_pair(self.eval_te, nt), nd, int(next_step["S"]), r2n, sch=self._sch_r2)
                    for _ in range(k):
                        next(g2n, None)
                self._r2_pre = dict(gen=g2n, res=r2n, k=k, S=int(next_step["S"]), noises_host=nh.clone(), noises_dev=nd, ev_noise=ev_noise,
                                    tokens_ori=tuple(t.clone() for t in nt))
        self._mark("classify_targets")
        # (FD_TAIL_REORDER=1 enqueues the recorded CLIP / DINO forward of the loss here, ahead of the tail's first read-back: measured slower)
        h_g = self.classify_begin(images, record=share)
        pre_g = None
        if share and self.use_img_loss and not _NO_TAIL_REORDER:
            small_g, fullbox_g = self.resize_small(images)
            pre_g = (fullbox_g,) + tuple(self.image_features(small_g, record=True))
        dv = self.device_tail
        rb = {}                # device_tail: what the step reports about itself, read back ONCE at the end (name -> device tensor)
        ind, boxes, per = self.classify_dev(h_g) if dv else self.classify_end(h_g)
        # ---- dynamic targets from the global batch (:1805-1837): gathered now, solved underneath R2, consumed by R3's loss
        self.start_dynamic_targets(per, B)
        out.update(images=images)
        (rb if dv else out).update(probs=per[0]["probs"], preds=per[0]["preds"])
        # ---- R2: images from the frozen original models (:1844-1858)
        if conc:
            self._mark("R2_tail_and_regularisers")
            cur.wait_event(ev_r2)              # everything downstream (classifier, feature encoders, loss) consumes images_ori on ``cur``
            images_ori.record_stream(cur)
        else:
            self._mark("R2_rollout")
            enc_ori = self.encode_pair(self.eval_te, tokens_ori) if (self.eval_te is not self.te or train_prefix or tokens_ori is not tokens) else enc
            lats = [self.rollout(self.eval_unet, enc_ori, noises[j:j + vb], S)[0] for j in range(0, B, vb)]
            self._mark("R2_vae")
            images_ori = torch.cat([self.decode(x) for x in lats])
            self._mark("R2_classify_regularisers")
        if conc and r2side is not None:
            h_o, clip_ori, dino_ori, face_ori = r2side["h_o"], r2side.get("clip_ori"), r2side.get("dino_ori"), r2side.get("face_ori")
            ind_o, boxes_o, per_o = self.classify_dev(h_o) if dv else self.classify_end(h_o)
        else:
            h_o = self.classify_begin(images_ori)
            if _NO_TAIL_REORDER:
                ind_o, boxes_o, per_o = self.classify_dev(h_o) if dv else self.classify_end(h_o)
            ind_o = h_o["ind"]
            if self.use_img_loss:                                                    # :1860-1862
                e_co, e_do = self.image_features(self.resize_small(images_ori)[0])
                clip_ori, dino_ori = F.normalize(e_co, dim=-1), F.normalize(e_do, dim=-1)
            if self.use_face_loss:                                                   # :1870
                from .sfnet import face_features
                ch_o, idx_o, _ = self.aligned_faces(images_ori, ind_o)
                face_ori = torch.zeros((B, 512), dtype=F32, device=dev)
                if len(idx_o):
                    face_ori[idx_o.long()] = F.normalize(face_features(self.face_net, ch_o)[0], dim=-1)
            if not _NO_TAIL_REORDER:                 # the read-back of R2's logits comes after its feature encoders have been enqueued
                ind_o, boxes_o, per_o = self.classify_dev(h_o) if dv else self.classify_end(h_o)
        out.update(images_ori=images_ori)
        (rb if dv else out).update(preds_ori=per_o[0]["preds"], probs_ori=per_o[0]["probs"])
        tgt = self.finish_dynamic_targets()
        targets = tgt[0][0]
        if dv:
            rb.update(targets=targets, uncertainty=tgt[0][1])
        else:
            out.update(targets=targets, uncertainty=tgt[0][1], targets_by_attr={a["name"]: t for a, (t, _) in zip(per, tgt)})
        # ---- R3: rollout with gradient (:1889-1933), all micro-batches at once with weights 1/n_j
        w, N_backward = microbatch_weights(B, args.train_GPU_batch_size)
        if share:
            (enc_g, inputs, ctxs), images_g, ind_g, boxes_g, per_g = shared, images, ind, boxes, per
        else:
            self._mark("R3_fwd_rollout")
            enc_g = self.encode_pair(self.te, tokens, record=rec_te, prefix=pv)
            x_final, inputs, ctxs = self.rollout(self.unet, enc_g, noises, S, keep_inputs=True, record_prompt=True,
                                                 keep_activations=self.keep_activations)
            self._mark("R3_fwd_vae")
            images_g = self.decode(x_final, record=True)
            ind_g, boxes_g, per_g = self.classify_dev(self.classify_begin(images_g, record=True)) if dv else self.classify(images_g, record=True)
        self._mark("R3_loss_and_image_grad")
        loss_by_attr = {}
        if dv:
            name, c0, k = self.attrs[0]
            ind_dev, w_dev = per_g[0]["ind_dev"], _h2d(w, dev)
            lf, dl = FD.fair_loss_and_grad(per_g[0]["logits"], targets, ind_dev, w_dev)
            dlog_full = torch.zeros((B, self.clf.num_classes), dtype=F32, device=dev)
            dlog_full[:, c0:c0 + k] = dl
            rb["loss_fair"] = lf
            out.update(images_grad=images_g, N_backward=N_backward)
        else:
            dlog_full = torch.zeros((B, self.clf.num_classes), dtype=F32)
            for (name, c0, k), a, (t_a, _) in zip(self.attrs, per_g, tgt):      # loss_ij = sum over attributes (:1932; exp-3 :2146)
                lf, dl = fair_loss_and_grad(a["logits"], t_a, ind_g, w)
                loss_by_attr[name] = lf
                dlog_full[:, c0:c0 + k] = dl
            loss_fair = loss_by_attr[self.attrs[0][0]]
            out.update(loss_fair=loss_fair, loss_fair_by_attr=loss_by_attr, images_grad=images_g, N_backward=N_backward)
        sel = ind_g.nonzero().view(-1)
        Himg, Wimg = images_g.shape[2], images_g.shape[3]
        d_img = None
        deferred = []          # host read-backs of reported values: executed once the whole backward has been enqueued
        if self.use_img_loss:
            # image-semantics term (:1904-1910, :1931-1932): w_i = (1/n_j) * weight_loss_img * dynamic_weight_i
            if share and pre_g is not None:
                fullbox, e_c, e_d = pre_g            # enqueued ahead of the tail's first read-back (see classify_targets above)
            else:
                small, fullbox = self.resize_small(images_g)
                e_c, e_d = self.image_features(small, record=True)
            self._fine("L_a_clip_dino_fwd_enqueued")
            tl, pl = [t for t, _ in tgt], [a["preds"] for a in per_o]
            if dv:
                dyn = FD.dynamic_weights(ind_dev, targets, per_o[0]["preds"], self.factors1[0])
                wi = w_dev * args.weight_loss_img * dyn
            else:
                if len(tl) == 1:
                    dyn = gen_dynamic_weights(ind_g, targets, per_o[0]["preds"], factor=self.factors1[0])
                else:
                    dyn = gen_dynamic_weights_multi(ind_g, tl, pl, self.factors1)
                wi = _h2d(w * args.weight_loss_img * dyn, dev)
            loss_clip, de_c = feature_loss_and_grad(e_c, clip_ori, wi)
            loss_dino, de_d = feature_loss_and_grad(e_d, dino_ori, wi)
            am = torch.stack([de_c.abs().max(), de_d.abs().max()]).float()
            if _HOST_SCALES:
                am = am.cpu()        # ONE read-back for both scales
                self._fine("L_b_after_amax_readback")
                dsmall = self.clip.backward(de_c, _pow2_scale(float(am[0]), 1.0))
                self.dino.backward(de_d, _pow2_scale(float(am[1]), 1.0), out=dsmall)
            else:
                # power-of-two scales chosen on the device: the scaled gradient enters with gscale = 1 and the result is un-scaled by the
                # exact inverse (a power-of-two multiply commutes with every rounding: same bits as the host-scale path)
                sc = _pow2_scale_dev(am, 1.0)
                dsmall = self.clip.backward(de_c * sc[0], 1.0).mul_(1.0 / sc[0])
                dsmall.addcmul_(self.dino.backward(de_d * sc[1], 1.0), 1.0 / sc[1])
            d_img = ops.crop_resize_bwd(dsmall, fullbox, B, Himg, Wimg, args.img_size_small)
            # apply_grad_hook_face (:1904, :1584-1617) acts on this path only: the classifier saw the un-hooked images
            if dv:       # the rectangle is a function of the two boxes (host); the factor of targets / original predictions (device)
                zt = torch.zeros(B, dtype=torch.long)
                rects, _ = face_grad_factors(boxes_g, boxes_o, zt, zt, 1.0, Himg, Wimg)
                has_box = _h2d(~(boxes_g == -1).all(dim=1), dev)
                ops.rect_scale(d_img, _h2d(rects, dev), FD.hook_factors(has_box, targets, per_o[0]["preds"], self.factors2[0]))
                rb.update(loss_CLIP=loss_clip.float(), loss_DINO=loss_dino.float(), dynamic_weights=dyn)
            else:
                if len(tl) == 1:
                    rects, facs = face_grad_factors(boxes_g, boxes_o, targets, per_o[0]["preds"], self.factors2[0], Himg, Wimg)
                else:
                    rects, facs = face_grad_factors_multi(boxes_g, boxes_o, tl, pl, self.factors2, Himg, Wimg)
                ops.rect_scale(d_img, _h2d(rects, dev), _h2d(facs, dev))
                # (the per-image regulariser values are only reported: they are read back at the end of the step, not here)
                deferred.append(lambda lc=loss_clip, ld=loss_dino: out.update(
                    loss_CLIP=lc.float().cpu(), loss_DINO=ld.float().cpu(), dynamic_weights=dyn,
                    loss=sum(loss_by_attr.values()) + args.weight_loss_img * dyn * (lc.float().cpu() + ld.float().cpu())))
            self._fine("L_c_clip_dino_bwd_enqueued")
        if self.use_face_loss:
            # face-realism term (:1917-1932): target = the original image's own face features when the target class equals the
            # original prediction with confidence >= face_gender_confidence_level, else the nearest database face
            from .sfnet import face_features, face_features_backward
            # exp-1 searches only for images with a target (:1926); the multi-attribute scripts search for every face (exp-3 :2135)
            if dv:
                # which faces carry a target is only known on the device: every face goes through the face network and the ones without a
                # target get weight 0 (same loss and gradient: their term is multiplied out; the reference skips them, :1926)
                from_ori = ind_dev & (targets != -1) & (targets == per_o[0]["preds"]) & (per_o[0]["probs"].max(dim=-1).values >= self.face_conf)
                has_d = ind_dev & (targets != -1)
                has = ind_g
                loss_face = torch.full((B,), -1.0, dtype=F32, device=dev)
            else:
                from_ori = ind_g.clone()
                for (t_a, _), a in zip(tgt, per_o):
                    from_ori &= (t_a != -1) & (t_a == a["preds"]) & (a["probs"].max(dim=-1).values >= self.face_conf)
                has = (ind_g & (targets != -1)) if len(tgt) == 1 else ind_g.clone()
                loss_face = torch.full((B,), -1.0)
            rows = has.nonzero().view(-1)
            if len(rows):
                chips_f, idx_f, A_f = self.aligned_faces(images_g, has)
                self._fine("L_c1_aligned_faces")
                feats, fctx = face_features(self.face_net, chips_f, record=True)
                self._fine("L_c2_sfnet_fwd_enqueued")
                fn = F.normalize(feats, dim=-1)
                tgt = self.nearest_face_feats(fn)
                rows_d = _h2d(rows, dev)
                use_ori = from_ori[rows_d] if dv else _h2d(from_ori[rows], dev)
                tgt = torch.where(use_ori[:, None], face_ori[rows_d], tgt)
                wf = (w_dev[rows_d] * args.weight_loss_face * has_d[rows_d]) if dv else _h2d(w[rows] * args.weight_loss_face, dev)
                lf_rows, df = feature_loss_and_grad(feats, tgt, wf)
                self._fine("L_c3_nearest_and_loss_enqueued")
                if dv:
                    loss_face[rows_d] = torch.where(has_d[rows_d], lf_rows.float(), torch.full_like(lf_rows, -1.0, dtype=F32))
                else:
                    deferred.append(lambda lf_rows=lf_rows, rows=rows: loss_face.__setitem__(rows, lf_rows.float().cpu()))
                if _HOST_SCALES:
                    dfmax = float(df.abs().max())
                    self._fine("L_c4_df_amax_readback")
                    dch = face_features_backward(self.face_net, fctx, df, _pow2_scale(dfmax, 1.0))
                else:
                    sf = _pow2_scale_dev(df.abs().max(), 1.0)
                    dch = face_features_backward(self.face_net, fctx, df * sf, 1.0).mul_(1.0 / sf)
                self._fine("L_c5_sfnet_bwd_enqueued")
                if d_img is None:
                    d_img = torch.zeros((B, 3, Himg, Wimg), dtype=F32, device=dev)
                ops.warp_affine_bwd(dch.contiguous(), idx_f, A_f, d_img, args.size_aligned_face)   # un-hooked images (:1901)
            if dv:
                rb["loss_face"] = loss_face
            else:
                deferred.append(lambda: out.update(loss_face=loss_face, loss=out.get("loss", sum(loss_by_attr.values())) + args.weight_loss_face * loss_face))
            self._fine("L_d_face_branch_done")
        # (device_tail: whether any logit gradient is non-zero is not known to the host -- the classifier backward runs whenever there is a face)
        any_dlog = bool(len(sel)) and (dv or float(dlog_full.abs().sum()) > 0)
        if any_dlog or d_img is not None:
            if any_dlog:
                dlog = dlog_full[_h2d(sel, dev)] if dv else _h2d(dlog_full[sel], dev)
                dchips = self.clf.backward(dlog.contiguous(), self.clf_gscale)
                full = dchips
                if len(sel) != B:
                    full = torch.zeros((B,) + tuple(dchips.shape[1:]), dtype=F32, device=dev)
                    full[_h2d(sel, dev)] = dchips
                bx = boxes_g.clone()
                bx[~ind_g] = 0
                d_fair = ops.crop_resize_bwd(full.contiguous(), _h2d(bx, dev), B, Himg, Wimg, args.size_face)
                d_img = d_fair if d_img is None else d_img.add_(d_fair)
            else:
                self.clf._ctx = None
            self._mark("R3_bwd_vae")
            coefs = self.sch.grad_coefs() * self.sch.chain_coefs()   # hook (:1128) x scheduler recurrence (:1131)
            gs = args.guidance_scale
            if _HOST_SCALES:
                vscale = _pow2_scale(float(d_img.abs().max()), 64.0)
                dz = self.vae.backward_images(d_img, vscale)
                g = dz * (1.0 / self.vae.config.scaling_factor)          # dL/dx_final  [B,4,h,w] fp32
                gscale = _pow2_scale(float(g.abs().max()) * float(abs(coefs).max()) * max(abs(gs), abs(1 - gs)), 64.0)
                step_scale, inv_gscale = [float(c * gscale) for c in coefs], None
            else:
                vs = _pow2_scale_dev(d_img.abs().max(), 64.0)
                dz = self.vae.backward_images(d_img * vs, 1.0)
                g = dz * ((1.0 / self.vae.config.scaling_factor) / vs)   # dL/dx_final  [B,4,h,w] fp32
                # the U-Net backward's scale stays on the device too: the timesteps' upstream gradients are multiplied by (c_i * gscale) as device
                # scalars, the kernels run with gscale = 1 and the LoRA gradients are un-scaled ONCE, after the per-stream buffers have been summed
                gscale_dev = _pow2_scale_dev(g.abs().max() * (float(abs(coefs).max()) * max(abs(gs), abs(1 - gs))), 64.0)
                step_scale = _h2d(torch.as_tensor(coefs, dtype=F32), dev) * gscale_dev
                inv_gscale, gscale = 1.0 / gscale_dev, 1.0
            out.update(g=g, coefs=coefs, gscale=gscale)
            self._mark("R3_bwd_unet")
            if train_unet or train_te or train_prefix:
                # The S per-timestep backwards are independent (the U-Net input is detached at every step, :1115): they are dealt round-robin
                # to ``bwd_streams`` HIP streams, each side stream accumulating its LoRA gradients into its own buffer, so that the many
                # launches which cannot fill the chip alone overlap with a neighbouring timestep's.  Shared cross-attention dK/dV: fp32 atomics.
                cur = torch.cuda.current_stream()
                self.unet.prepare_backward()         # lazily built weight copies exist before any side stream can read them
                if not _ATOMIC_DKDV:
                    self.unet.prepare_backward_slots(S)   # per-timestep dK / dV pairs of the shared cross-attention K / V: no atomics, fixed-order sum
                nst = max(1, min(self.bwd_streams, S)) if self.concurrent_bwd else 1
                sides = [self._side_stream(k) for k in range(1, nst)]
                virtual = self.bwd_virtual
                graveyard = [] if self.bwd_keep_alive else None
                # upstream gradient of the CFG pair eps = eps_u + gs (eps_c - eps_u): [(1 - gs) g ; gs g], built once; a timestep scales it.
                # Built on the launch stream BEFORE the side streams take their dependency on it: enqueued after ``wait_stream`` (as it was for
                # most of round 3) the side streams' first timesteps could read it while its ``cat`` was still running -- a race that showed
                # as a 4e-4 schedule-to-schedule gradient difference (once as NaN) whenever kernel timing shifted.
                gpair = torch.cat([g * (1.0 - gs), g * gs])
                for k, side in enumerate(sides, 1):
                    for bank in self.banks:
                        bank.grad_alt(k).zero_()
                    side.wait_stream(cur)
                for i in range(S):
                    k = i % nst
                    on_side = k > 0
                    side = sides[k - 1] if on_side else None
                    with (torch.cuda.stream(side) if (on_side and not virtual) else contextlib.nullcontext()):
                        if i in ctxs:
                            self.unet._ctx = ctxs.pop(i)        # activations kept from the forward rollout
                            if graveyard is not None:
                                graveyard.append(_all_tensors(self.unet._ctx))
                            if on_side and not virtual:
                                _record_stream(self.unet._ctx, side)
                        else:                                   # gradient-checkpointed recompute of this timestep
                            x = ops.to_f16(inputs[i])
                            self.unet.forward_step(x if _CFG_PAIR else x.repeat(2, 1, 1, 1), i, record=True, pair=_CFG_PAIR)
                        for bank in self.banks:
                            bank.accum = bank.grad_alt(k) if on_side else bank.grad
                        unet_mod.BWD_SLOT[0] = i
                        self.unet.backward_step(gpair * step_scale[i], gscale)
                unet_mod.BWD_SLOT[0] = None
                for bank in self.banks:
                    bank.accum = bank.grad
                if self.debug_partials is not None:
                    torch.cuda.synchronize()
                    self.debug_partials.append([self.banks[0].grad.clone()] + [self.banks[0].grad_alt(k).clone() for k in range(1, nst)])
                for k, side in enumerate(sides, 1):
                    cur.wait_stream(side)
                    for bank in self.banks:
                        bank.grad.add_(bank.grad_alt(k))
                graveyard = None
                denc = self.unet.finish_prompt_backward(gscale, need_denc=rec_te)
                if rec_te:
                    L = enc_g.shape[1]
                    dx0 = self.te.backward(denc.view(2, L, -1), gscale)
                    if train_prefix:     # the prefix vectors sit at positions 1..n of the PROMPT row (row 1; row 0 is the uncond sequence)
                        n = self.prefix.n
                        self.prefix.bank.grad_view("token_embedding.weight")[1:].add_(dx0[1, 1:1 + n].float(), alpha=1.0 / gscale)
                if inv_gscale is not None:      # device-side loss scale: every bank's gradient of this step carries it exactly once
                    for bank in self.banks:
                        bank.grad.mul_(inv_gscale)
        else:
            self.vae._ctx = self.clf._ctx = None
        ctxs.clear()
        if self._r2_pre is not None and self.r2_prefetch_late > 0:
            k2 = max(0, min(self.r2_prefetch_late, self._r2_pre["S"] - self._r2_pre["k"]))
            with torch.cuda.stream(self._side_stream("r2")):
                for _ in range(k2):
                    next(self._r2_pre["gen"], None)
            self._r2_pre["k"] += k2
        for fn in deferred:
            fn()
        # ---- gradient sync, guard, update (:1998-2029)
        self._mark("sync_update")
        if dv:       # ONE read-back for everything the step reports, taken together with the finite flag inside sync_and_update
            keys = sorted(rb)
            self._pending_readback = torch.cat([rb[k].reshape(-1).to(F32) for k in keys]) if keys else None
        out["grad_is_finite"] = self.sync_and_update(N_backward)
        if dv:
            host = self._readback_host if self._pending_readback is None else self._pending_readback.cpu()    # (a replaced sync_and_update: read here)
            self._pending_readback = None
            off = 0
            for k in keys:
                n = rb[k].numel()
                v = host[off:off + n].reshape(rb[k].shape)
                off += n
                out[k] = v.long() if rb[k].dtype == torch.long else v
            name = self.attrs[0][0]
            out.update(targets_by_attr={name: out["targets"]}, loss_fair_by_attr={name: out["loss_fair"]}, loss=out["loss_fair"].clone())
            if "loss_CLIP" in out:
                out["loss"] = out["loss"] + args.weight_loss_img * out["dynamic_weights"] * (out["loss_CLIP"] + out["loss_DINO"])
            if "loss_face" in out:
                out["loss"] = out["loss"] + args.weight_loss_face * out["loss_face"]
        self._mark("end")
        return out

    def sync_and_update(self, N_backward, apply=True):
        args = self.args
        self.flag.zero_()
        self.allreduce_grads()
        for bank in self.banks:
            ops.grad_finite_scale(bank.grad, 1.0 / (self.world * N_backward), self.flag)
        pend, self._pending_readback = getattr(self, "_pending_readback", None), None
        if pend is not None:      # the step's reported values ride the flag's read-back: one device -> host copy per step
            host = torch.cat([self.flag.to(F32), pend]).cpu()
            finite, self._readback_host = int(host[0]) == 0, host[1:]
        else:
            finite = int(self.flag.item()) == 0  # checked after the all-reduce so every rank takes the same branch
        self.last_lr = args.learning_rate * lr_lambda(getattr(args, "lr_scheduler", "constant"), self.lr_step,
                                                      getattr(args, "lr_warmup_steps", 0), getattr(args, "max_train_steps", 1),
                                                      getattr(args, "lr_num_cycles", 1), getattr(args, "lr_power", 1.0), args.learning_rate)
        if apply:
            self.lr_step += 1
        if finite and apply:
            self.opt_step += 1
            for bank, ema in zip(self.banks, self.ema):
                ops.adamw_ema(bank.flat, bank.grad, bank.exp_avg, bank.exp_avg_sq, bank.ema, self.last_lr, args.adam_beta1,
                              args.adam_beta2, args.adam_epsilon, args.adam_weight_decay, self.opt_step, ema.next_one_minus_decay())
            if getattr(args, "train_unet", False):
                self.unet.refresh_lora()
            if getattr(args, "train_text_encoder", False):
                self.te.refresh_lora()
        return finite
