"""CPU fp32 restatement of the reference's fairness step (TEST ORACLE).

Every function cites the reference lines it follows; all citations are into
/root/reference/exp-1-debias-gender/1-main-debias.py unless another file is named.
Pure functions here are PINNED by tests/golden/reference_pure_functions.json
(produced by executing the reference's own source, tests/golden/make_golden.py).
The rollout/backward uses plain torch autograd exactly the way the reference does
(detach at the U-Net input, per-step grad hooks), so it is the ground truth for
the product's scalar-chain recompute-backward.
"""
import itertools
import math

import numpy as np
import scipy.stats
import torch
import torch.nn.functional as F

from .nn_sfnet import SRC_LANDMARKS, get_face_feats, image_pipeline, semantic_search
from .nn_vit import CLIP_IMAGE_MEAN, CLIP_IMAGE_STD, DINO_IMAGE_MEAN, DINO_IMAGE_STD, image_features


# --------------------------------------------------------------------------- helpers
def make_grad_hook(coef):  # :219-220
    return lambda x: coef * x


def expand_bbox(bbox, expand_coef, target_ratio):  # :238-265
    bw, bh = bbox[2] - bbox[0], bbox[3] - bbox[1]
    cur = bh / bw
    if cur > target_ratio:
        more_h = bh * expand_coef
        more_w = (bh + more_h) / target_ratio - bw
    else:
        more_w = bw * expand_coef
        more_h = (bw + more_w) * target_ratio - bh
    return [int(round(bbox[0] - more_w * 0.5)), int(round(bbox[1] - more_h * 0.5)),
            int(round(bbox[2] + more_w * 0.5)), int(round(bbox[3] + more_h * 0.5))]


def crop_face(img, bbox, target_size, fill_value):  # :267-290
    """img [3,H,W]; torchvision Pad(constant)+Resize(bilinear, no antialias on tensors)."""
    H, W = img.shape[-2:]
    l, r = max(bbox[0], 0), min(bbox[2], W)
    b, t = max(bbox[1], 0), min(bbox[3], H)
    pl, pr = max(-bbox[0], 0), max(-(W - bbox[2]), 0)
    pt, pb = max(-bbox[1], 0), max(-(H - bbox[3]), 0)
    face = img[:, b:t, l:r]
    if pl > 0 or pt > 0 or pr > 0 or pb > 0:
        face = F.pad(face, [pl, pr, pt, pb], value=fill_value)
    return F.interpolate(face[None], size=list(target_size), mode="bilinear", align_corners=False, antialias=False)[0]


class SyntheticFaceProvider:
    """Stand-in for the insightface/dlib detector seam (:1192-1353):
    every image has one face whose raw detector box is the centred half-size square;
    the reference's own expand_bbox(0.5, 1) + crop_face then apply."""

    def __init__(self, size_face=224):
        self.size_face = size_face

    def raw_box(self, H, W):
        return [0.25 * W, 0.25 * H, 0.75 * W, 0.75 * H]

    def __call__(self, images, fill_value=-1):
        N, _, H, W = images.shape
        bbox = expand_bbox(self.raw_box(H, W), 0.5, 1)
        ind = torch.ones(N, dtype=torch.bool)
        boxes = torch.tensor([bbox] * N, dtype=torch.long)
        chips = torch.stack([crop_face(images[i], bbox, [self.size_face, self.size_face], fill_value) for i in range(N)])
        return ind, boxes, chips

    def landmarks(self, images):
        """The 5-point template of the aligned chip scaled into the raw detector box."""
        N, _, H, W = images.shape
        pts = SRC_LANDMARKS / 112.0 * np.array([0.5 * W, 0.5 * H]) + np.array([0.25 * W, 0.25 * H])
        return [pts.copy() for _ in range(N)]

    def aligned(self, images, crop=112):
        """aligned_face_chips of get_face (:1337-1338): image_pipeline on every image (differentiable w.r.t. the image)."""
        return torch.stack([image_pipeline(images[i], lm, crop) for i, lm in enumerate(self.landmarks(images))])


def get_face_gender(classifier, face_chips, selector=None, fill_value=-1, slice_fn=None):  # :1355-1401
    x = face_chips[selector] if selector is not None else face_chips
    if x.shape[0] == 0:
        logits_g = torch.empty([0, 2], dtype=face_chips.dtype)
        probs = torch.empty([0, 2], dtype=face_chips.dtype)
        preds = torch.empty([0], dtype=torch.int64)
    else:
        logits = classifier(x)
        logits_g = logits.view([logits.shape[0], -1, 2])[:, 20, :] if slice_fn is None else slice_fn(logits)
        probs = torch.softmax(logits_g, dim=-1)
        preds = probs.max(dim=-1).indices
    if selector is None:
        return preds, probs, logits_g
    def scatter(v):
        new = torch.ones([selector.shape[0]] + list(v.shape[1:]), dtype=v.dtype) * fill_value
        new[selector] = v
        return new
    return scatter(preds), scatter(probs), scatter(logits_g)


@torch.no_grad()
def generate_dynamic_targets(probs, target_ratio=0.5, w_uncertainty=False):  # :1403-1447
    idxs = (probs != -1).all(dim=-1)
    p = probs[idxs]
    rank = torch.argsort(torch.argsort(p[:, 1]))
    targets = (rank >= (rank.shape[0] * target_ratio)).long()
    targets_all = torch.ones([probs.shape[0]], dtype=torch.long) * (-1)
    targets_all[idxs] = targets
    if not w_uncertainty:
        return targets_all
    unc = torch.ones([p.shape[0]], dtype=probs.dtype) * (-1)
    unc[targets == 1] = torch.tensor(1 - scipy.stats.binom.cdf(rank[targets == 1].numpy(), p.shape[0], 1 - target_ratio)).to(probs.dtype)
    unc[targets == 0] = torch.tensor(scipy.stats.binom.cdf(rank[targets == 0].numpy(), p.shape[0], target_ratio)).to(probs.dtype)
    unc_all = torch.ones([probs.shape[0]], dtype=probs.dtype) * (-1)
    unc_all[idxs] = unc
    return targets_all, unc_all


@torch.no_grad()
def generate_dynamic_targets_multi(probs_list, class_cdfs, num_samples=100, generator=None, age_asymmetric=False):
    """exp-3-debias-gender-race/1-main-debias.py:1459-1569 and exp-4-debias-gender-race-age/1-main-debias.py:1477-1615
    restated with the transport problem solved as a *linear program* (scipy HiGHS) -- ``ot.emd`` (POT 0.9.3,
    environment.yml:172) is not installable here, so this piece is PARITY UNPINNED; it is an independent solver from the
    product's assignment formulation, compared on the averaged plan as SURVEY.md 8c prescribes."""
    import itertools as it
    from scipy.optimize import linprog
    n = probs_list[0].shape[0]
    sizes = [p.shape[1] for p in probs_list]
    idx = torch.ones(n, dtype=torch.bool)
    for p in probs_list:
        idx &= (p != -1).all(dim=-1)
    N = int(idx.sum())
    res_t = [torch.full([n], -1, dtype=torch.long) for _ in sizes]
    res_u = [torch.full([n], -1.0) for _ in sizes]
    if N == 0:
        return list(zip(res_t, res_u))
    P = [p[idx].double().numpy() for p in probs_list]
    cells = list(it.product(*[range(k) for k in sizes]))
    K = len(cells)
    draws = []
    for cdf in class_cdfs:
        u = torch.rand([num_samples, N], generator=generator)
        cls = torch.zeros_like(u, dtype=torch.long)
        lo = 0.0
        for c, hi in enumerate(cdf):
            if c > 0:
                cls[(u > lo) & (u <= hi)] = c
            lo = hi
        draws.append(cls.numpy())
    M = np.zeros((N, K))
    for j, cell in enumerate(cells):
        for i in range(N):
            sq = 0.0
            for a, c in enumerate(cell):
                tgt = np.zeros(sizes[a]); tgt[c] = 1.0
                if age_asymmetric and a == len(sizes) - 1 and c == 1:
                    sq += ((P[a][i][0] - 0) * 2) ** 2 + (P[a][i][1] - 1) ** 2
                else:
                    sq += np.linalg.norm(P[a][i] - tgt) ** 2
            M[i, j] = sq ** 0.5
    # equality constraints: rows sum to 1, columns sum to the drawn counts
    A_eq = np.zeros((N + K, N * K))
    for i in range(N):
        A_eq[i, i * K:(i + 1) * K] = 1
    for j in range(K):
        A_eq[N + j, j::K] = 1
    tp = np.zeros((N, K))
    for s_ in range(num_samples):
        counts = np.zeros(K)
        for i in range(N):
            j = 0
            for a in range(len(sizes)):
                j = j * sizes[a] + draws[a][s_][i]
            counts[j] += 1
        r = linprog(M.reshape(-1), A_eq=A_eq, b_eq=np.concatenate([np.ones(N), counts]), bounds=(0, None), method="highs")
        tp += r.x.reshape(N, K)
    tp = torch.tensor(tp / tp[0].sum(), dtype=torch.float32)
    for a in range(len(sizes)):
        marg = torch.zeros(N, sizes[a])
        for j, cell in enumerate(cells):
            marg[:, cell[a]] += tp[:, j]
        res_t[a][idx] = marg.argmax(dim=-1)
        res_u[a][idx] = 1 - marg.max(dim=-1).values
    return list(zip(res_t, res_u)), tp


def apply_grad_hook_face(images, face_bboxs, face_bboxs_ori, targets, preds_ori, factor=0.1):  # :1584-1617
    out = []
    for image, bb, bbo, target, pred_ori in itertools.zip_longest(images, face_bboxs, face_bboxs_ori, targets, preds_ori):
        if (bb == -1).all():
            out.append(image[None])
            continue
        img_w, img_h = image.shape[1:]  # (H, W) swapped in the reference, harmless for squares (:1592)
        l, r = max(bb[0], bbo[0], 0), min(bb[2], bbo[2], img_w)
        b, t = max(bb[1], bbo[1], 0), min(bb[3], bbo[3], img_h)
        face = image[:, b:t, l:r].clone()
        coef = 1 if (target != -1 and target == pred_ori) else factor
        face.register_hook(make_grad_hook(coef))
        add = torch.zeros_like(image)
        add[:, b:t, l:r] = face
        mask = torch.zeros_like(image)
        mask[:, b:t, l:r] = 1
        out.append((mask * add + (1 - mask) * image)[None])
    return torch.cat(out)


def gen_dynamic_weights(face_indicators, targets, preds_ori, factor=0.2):  # :1619-1633
    w = []
    for ind, target, pred_ori in itertools.zip_longest(face_indicators, targets, preds_ori):
        if not bool(ind):
            w.append(1)
        elif target == -1:
            w.append(factor)
        elif target == pred_ori:
            w.append(1)
        else:
            w.append(factor)
    return torch.tensor(w, dtype=torch.float32)


def grad_coefs(scheduler):  # :1105-1109
    c = []
    for t in scheduler.timesteps:
        acp = scheduler.alphas_cumprod[t]
        c.append(acp.sqrt().item() * (1 - acp).sqrt().item() / (1 - scheduler.alphas[t].item()))
    c = np.array(c)
    return c / (math.prod(c) ** (1 / len(c)))


# --------------------------------------------------------------------------- rollouts
def encode_prompts(text_encoder, prompt_ids, prompt_mask, uncond_ids, uncond_mask, N):
    """:1007-1036 / :1074-1102 with tokenisation replaced by explicit ids (no vocab here):
    prompt ids/mask [L], uncond ids/mask [L] -> [2N, L, D] (negative first)."""
    pe = text_encoder(prompt_ids[None].repeat(N, 1), prompt_mask[None].repeat(N, 1))[0]
    ne = text_encoder(uncond_ids[None].repeat(N, 1), uncond_mask[None].repeat(N, 1))[0]
    return torch.cat([ne, pe])


def prefix_encoders(text_encoder, prompt_ids, fair_ids, fair_table, eos_id, plain_tokens):
    """exp-2-debias-gender-token/1-main-debias.py:1049-1113 (= :1142-1210 for the gradient rollout): the two prompt encodings of a
    prefix-tuning step as closures ``N -> [2N, L, D]`` (negative first).
    * finetuned side (``which_prefix_embedding=prefix_embedding``, :1061-1086): ``prompt_ids`` already carries the n prefix-token ids after
      BOS; FairEmbeddings + ``text_model_forward`` with the all-ones mask; negative embeddings as ``pipe._encode_prompt`` builds them
      (``[""]`` padded to the prompt length, ``attention_mask=None``);
    * original side (``which_prefix_embedding=None``, :1087-1112): the plain prompt with its mask, the empty prompt WITHOUT a mask (:1107-1110).
    ``fair_table`` [n+1, D] may require grad: it is the only trained tensor (:946)."""
    L = prompt_ids.shape[0]

    def encode(N):
        pe = text_encoder(prompt_ids[None].repeat(N, 1), torch.ones(N, L, dtype=torch.long), fair=(fair_ids, fair_table))[0]
        uids = torch.cat([prompt_ids[:1], torch.full((L - 1,), eos_id, dtype=prompt_ids.dtype)])
        ne = text_encoder(uids[None].repeat(N, 1), None)[0]
        return torch.cat([ne, pe])

    def encode_ori(N):
        pid, pm, uid, _ = plain_tokens
        pe = text_encoder(pid[None].repeat(N, 1), pm[None].repeat(N, 1))[0]
        ne = text_encoder(uid[None].repeat(N, 1), None)[0]
        return torch.cat([ne, pe])
    return encode, encode_ori


@torch.no_grad()
def generate_image_no_gradient(tokens, noises, S, text_encoder, unet, vae, scheduler, guidance_scale=7.5, dtype=torch.float32,
                               trace=None, encode=None):  # :998-1061
    N = noises.shape[0]
    emb = (encode(N) if encode is not None else encode_prompts(text_encoder, *tokens, N)).to(dtype)
    scheduler.set_timesteps(S)
    latents = noises
    for i, t in enumerate(scheduler.timesteps):
        x = scheduler.scale_model_input(torch.cat([latents.to(dtype)] * 2), t)
        eps = unet(x, t, encoder_hidden_states=emb).sample.to(torch.float32)
        eu, ec = eps.chunk(2)
        eps = eu + guidance_scale * (ec - eu)
        latents = scheduler.step(eps, t, latents).prev_sample
        if trace is not None:
            trace.append(latents.clone())
    latents = 1 / vae.config.scaling_factor * latents
    return vae.decode(latents.to(vae.dtype)).sample.clamp(-1, 1)


@torch.no_grad()
def generate_image_w_prefix_embedding(prompt_ids, noises, fair_ids, fair_table, eos_id, S, text_encoder, unet, vae, scheduler, guidance_scale=7.5):
    """gen-images.py:273-343 restated: the prompt already carries the n prefix-token ids after BOS (``"".join(prefix_tokens) + prompt``,
    :526); prompt embeddings through FairEmbeddings + text_model_forward WITH the (all-ones) attention mask; negative embeddings as
    ``StableDiffusionPipeline._encode_prompt`` (diffusers 0.19.3) builds them: ``[""]`` padded to the prompt length with EOS, text encoder
    called with ``attention_mask=None`` (SD-v1.5's config has no ``use_attention_mask``)."""
    N, L = noises.shape[0], prompt_ids.shape[0]
    pe = text_encoder(prompt_ids[None].repeat(N, 1), torch.ones(N, L, dtype=torch.long), fair=(fair_ids, fair_table))[0]
    uids = torch.cat([prompt_ids[:1], torch.full((L - 1,), eos_id, dtype=prompt_ids.dtype)])
    ne = text_encoder(uids[None].repeat(N, 1), None)[0]
    emb = torch.cat([ne, pe])
    scheduler.set_timesteps(S)
    latents = noises
    for t in scheduler.timesteps:
        x = scheduler.scale_model_input(torch.cat([latents] * 2), t)
        eps = unet(x, t, encoder_hidden_states=emb).sample.to(torch.float32)
        eu, ec = eps.chunk(2)
        latents = scheduler.step(eu + guidance_scale * (ec - eu), t, latents).prev_sample
    return vae.decode(1 / vae.config.scaling_factor * latents).sample.clamp(-1, 1)


def generate_image_w_gradient(tokens, noises, S, text_encoder, unet, vae, scheduler, guidance_scale=7.5, dtype=torch.float32,
                              trace=None, encode=None):  # :1063-1136
    N = noises.shape[0]
    emb = (encode(N) if encode is not None else encode_prompts(text_encoder, *tokens, N)).to(dtype)
    scheduler.set_timesteps(S)
    coefs = grad_coefs(scheduler)
    latents = noises
    for i, t in enumerate(scheduler.timesteps):
        x = scheduler.scale_model_input(torch.cat([latents.detach().to(dtype)] * 2), t)
        eps = unet(x, t, encoder_hidden_states=emb).sample.to(torch.float32)
        eu, ec = eps.chunk(2)
        eps = eu + guidance_scale * (ec - eu)
        if eps.requires_grad:
            eps.register_hook(make_grad_hook(coefs[i]))
        latents = scheduler.step(eps, t, latents).prev_sample
        if trace is not None:
            trace.append(latents.detach().clone())
    latents = 1 / vae.config.scaling_factor * latents
    return vae.decode(latents.to(vae.dtype)).sample.clamp(-1, 1)


# --------------------------------------------------------------------------- EMA / step
class EMAModel:
    """diffusers==0.19.3 ``training_utils.EMAModel`` (defaults: no warm-up schedule,
    decay floor ``(1+n)/(10+n)``); reference :823/:874, stepped at :2025-2029."""

    def __init__(self, parameters, decay=0.9999):
        self.shadow_params = [p.clone().detach() for p in parameters]
        self.decay = decay
        self.optimization_step = 0

    def get_decay(self, optimization_step):
        step = max(0, optimization_step - 1)
        if step <= 0:
            return 0.0
        return max(min((1 + step) / (10 + step), self.decay), 0.0)

    @torch.no_grad()
    def step(self, parameters):
        self.optimization_step += 1
        omd = 1 - self.get_decay(self.optimization_step)
        for s, p in zip(self.shadow_params, parameters):
            s.sub_(omd * (s - p))


def fairness_step(models, tokens, noises, S, cfg, world=None, attrs=None, targets_by_attr=None):
    """One training step (:1746-2029) on one rank, synthetic face provider.
    loss_ij = loss_fair + weight_loss_img * dynamic_weights * (loss_CLIP + loss_DINO) (:1904-1932) when ``models`` holds the
    image encoders ``clip`` / ``dino`` (oracle.nn_vit) and cfg["weight_loss_img"] != 0, plus
    ``weight_loss_face * loss_face`` (:1917-1932) when it holds ``face_net`` (oracle.nn_sfnet.SFNet20) and ``face_db`` (normalised
    [M,512] database of FaceFeatsModel, :80-92) and cfg["weight_loss_face"] != 0.

    models: dict(text_encoder, unet, vae, classifier, scheduler, eval_text_encoder, eval_unet)
    cfg: dict(train_GPU_batch_size, val_GPU_batch_size, uncertainty_threshold, factor2, guidance_scale, size_face, slice_fn);
         exp-2 adds ``encode`` / ``encode_ori`` (``prefix_encoders``): how the finetuned and the original side embed the prompt
    world: optional (rank, world_size, probs_all) -- when given, the dynamic targets use the
           gathered ``probs_all`` of every rank (:1805-1837).
    Returns dict with images, probs, targets, uncertainty, loss_fair (per image, -1 sentinel),
    latents trace of R1, and N_backward; LoRA grads are left in the params' ``.grad``.
    """
    te, unet, vae, clf, sch = (models[k] for k in ("text_encoder", "unet", "vae", "classifier", "scheduler"))
    faces = cfg.get("face_provider") or SyntheticFaceProvider(cfg.get("size_face", 224))    # any get_face stand-in with the same returns
    gs, B = cfg.get("guidance_scale", 7.5), noises.shape[0]
    slice_fn = cfg.get("slice_fn")
    out = {}
    with torch.no_grad():
        trace = []
        vb = cfg["val_GPU_batch_size"]
        enc_fn, enc_ori_fn = cfg.get("encode"), cfg.get("encode_ori")
        images = torch.cat([generate_image_no_gradient(tokens, noises[j:j + vb], S, te, unet, vae, sch, gs,
                                                       trace=trace if j == 0 else None, encode=enc_fn) for j in range(0, B, vb)])
        ind, boxes, chips = faces(images)
        preds, probs, _ = get_face_gender(clf, chips, selector=ind, slice_fn=slice_fn)
        probs_all = probs if world is None else world[2]
        targets_all, unc_all = generate_dynamic_targets(probs_all, w_uncertainty=True)
        targets_all[unc_all > cfg["uncertainty_threshold"]] = -1
        r = 0 if world is None else world[0]
        targets, unc = targets_all[B * r:B * (r + 1)], unc_all[B * r:B * (r + 1)]
        images_ori = torch.cat([generate_image_no_gradient(tokens, noises[j:j + vb], S, models["eval_text_encoder"],
                                                           models["eval_unet"], vae, sch, gs, encode=enc_ori_fn) for j in range(0, B, vb)])
        ind_o, boxes_o, chips_o = faces(images_ori)
        preds_o, probs_o, _ = get_face_gender(clf, chips_o, selector=ind_o, slice_fn=slice_fn)
        w_img = cfg.get("weight_loss_img", 0.0) if ("clip" in models and "dino" in models) else 0.0
        if w_img:
            small_o = resize_small(images_ori, cfg.get("img_size_small", 224))              # :1860-1862
            clip_o = image_features(models["clip"], small_o, CLIP_IMAGE_MEAN, CLIP_IMAGE_STD)
            dino_o = image_features(models["dino"], small_o, DINO_IMAGE_MEAN, DINO_IMAGE_STD)
        w_face = cfg.get("weight_loss_face", 0.0) if ("face_net" in models and "face_db" in models) else 0.0
        crop = cfg.get("size_aligned_face", 112)
        if w_face:
            face_feats_o = get_face_feats(models["face_net"], faces.aligned(images_ori, crop))            # :1870
    out.update(images=images, images_ori=images_ori, probs=probs, preds=preds, targets=targets, uncertainty=unc,
               preds_ori=preds_o, probs_ori=probs_o, latents_trace=trace)
    tb = cfg["train_GPU_batch_size"]
    N_backward = math.ceil(B / tb)
    loss_fair = torch.ones(B) * (-1)
    loss_CLIP, loss_DINO, loss_all, loss_face = torch.ones(B) * (-1), torch.ones(B) * (-1), torch.ones(B) * (-1), torch.ones(B) * (-1)
    images_g = []
    for j in range(N_backward):
        idx = list(range(B))[j * tb:(j + 1) * tb]
        img = generate_image_w_gradient(tokens, noises[idx], S, te, unet, vae, sch, gs, encode=enc_fn)
        ind_j, boxes_j, chips_j = faces(img)
        preds_j, probs_j, logits_j = get_face_gender(clf, chips_j, selector=ind_j, slice_fn=slice_fn)
        img_raw = img
        img = apply_grad_hook_face(img, boxes_j, boxes_o[idx], targets[idx], preds_o[idx], factor=cfg["factor2"])
        # NB (:1904-1915): the hooked images feed only the CLIP/DINO terms in the reference; the
        # fairness loss uses logits computed from the un-hooked chips, so the hook does not touch it.
        lf = torch.ones(len(idx)) * (-1)
        w = ((ind_j == True) * (targets[idx] != -1)).nonzero().view([-1])  # noqa: E712
        lf[w] = F.cross_entropy(logits_j[w], targets[idx][w], reduction="none")
        loss_ij = lf
        if w_img:
            small = resize_small(img, cfg.get("img_size_small", 224))                       # :1905
            lc = 1 - (image_features(models["clip"], small, CLIP_IMAGE_MEAN, CLIP_IMAGE_STD) * clip_o[idx]).sum(dim=-1)
            ld = 1 - (image_features(models["dino"], small, DINO_IMAGE_MEAN, DINO_IMAGE_STD) * dino_o[idx]).sum(dim=-1)
            dyn = gen_dynamic_weights(ind_j, targets[idx], preds_o[idx], factor=cfg.get("factor1", 0.2))
            loss_ij = lf + w_img * dyn * (lc + ld)                                          # :1932 without the face term
            loss_CLIP[idx], loss_DINO[idx] = lc.detach(), ld.detach()
        if w_face:                                                                           # :1917-1929, :1932
            aligned_j = faces.aligned(img_raw, crop)        # get_face runs before the hook (:1901): un-hooked images
            t_j, p_o, pr_o = targets[idx], preds_o[idx], probs_o[idx]
            lface = torch.ones(len(idx)) * (-1)
            from_ori = ((ind_j == True) * (t_j != -1) * (t_j == p_o) *  # noqa: E712
                        (pr_o.max(dim=-1).values >= cfg.get("face_gender_confidence_level", 0.9))).nonzero().view([-1]).tolist()
            if len(from_ori) > 0:
                f1 = get_face_feats(models["face_net"], aligned_j[from_ori])
                lface[from_ori] = (1 - (f1 * face_feats_o[idx][from_ori]).sum(dim=-1)).to(lface.dtype)
            from_search = sorted(set(((ind_j == True) * (t_j != -1)).nonzero().view([-1]).tolist()) - set(from_ori))  # noqa: E712
            if len(from_search) > 0:
                f2 = get_face_feats(models["face_net"], aligned_j[from_search])
                lface[from_search] = (1 - (f2 * semantic_search(models["face_db"], f2)).sum(dim=-1)).to(lface.dtype)
            loss_ij = loss_ij + w_face * lface
            loss_face[idx] = lface.detach()
        if loss_ij.requires_grad:
            loss_ij.mean().backward()
        loss_fair[idx] = lf.detach()
        loss_all[idx] = loss_ij.detach()
        images_g.append(img.detach())
    out.update(loss_fair=loss_fair, loss_CLIP=loss_CLIP, loss_DINO=loss_DINO, loss_face=loss_face, loss=loss_all, N_backward=N_backward,
               images_grad=torch.cat(images_g))
    return out


def resize_small(images, size):
    """``transforms.Resize(size)`` on a square float tensor batch (torchvision 0.16: bilinear, antialias default "warn" = off)."""
    return F.interpolate(images, size=(size, size), mode="bilinear", align_corners=False)


def _mismatch_factor(ts, ps, factors):
    bad = [f for t, p, f in zip(ts, ps, factors) if t != p]
    return 1 if not bad else min(bad)


def gen_dynamic_weights_multi(face_indicators, targets_list, preds_ori_list, factors):  # exp-3 :1786-1803, exp-4 :1870-1895
    w = []
    for i, ind in enumerate(face_indicators):
        w.append(min(factors) if not bool(ind) else _mismatch_factor([t[i] for t in targets_list], [p[i] for p in preds_ori_list], factors))
    return torch.tensor(w, dtype=torch.float32)


def apply_grad_hook_face_multi(images, face_bboxs, face_bboxs_ori, targets_list, preds_ori_list, factors):  # exp-3 :1751-1783, exp-4 :1823-1867
    out = []
    for i, (image, bb, bbo) in enumerate(zip(images, face_bboxs, face_bboxs_ori)):
        if (bb == -1).all():
            out.append(image[None])
            continue
        img_w, img_h = image.shape[1:]
        l, r = max(bb[0], bbo[0], 0), min(bb[2], bbo[2], img_w)
        b, t = max(bb[1], bbo[1], 0), min(bb[3], bbo[3], img_h)
        face = image[:, b:t, l:r].clone()
        face.register_hook(make_grad_hook(_mismatch_factor([x[i] for x in targets_list], [x[i] for x in preds_ori_list], factors)))
        add = torch.zeros_like(image)
        add[:, b:t, l:r] = face
        mask = torch.zeros_like(image)
        mask[:, b:t, l:r] = 1
        out.append((mask * add + (1 - mask) * image)[None])
    return torch.cat(out)


def fairness_step_multi(models, tokens, noises, S, cfg, attrs, targets_by_attr):
    """Multi-attribute variant of the R3 part (exp-3 `:2079-2155`): loss_ij = sum_a CE_a with -1 sentinels; the dynamic
    targets are passed in (they come from a Monte-Carlo OT procedure, tested separately).  attrs: [(name, col0, width)].
    With ``clip``/``dino`` and/or ``face_net``/``face_db`` in ``models`` (and ``eval_unet``/``eval_text_encoder`` for R2) the regulariser
    terms of exp-3 `:2106-2147` are added: cfg factors1 / factors2 (per attribute), weight_loss_img, weight_loss_face, face_conf."""
    te, unet, vae, clf, sch = (models[k] for k in ("text_encoder", "unet", "vae", "classifier", "scheduler"))
    faces = SyntheticFaceProvider(cfg.get("size_face", 224))
    gs, B, tb = cfg.get("guidance_scale", 7.5), noises.shape[0], cfg["train_GPU_batch_size"]
    N_backward = math.ceil(B / tb)
    losses = {a[0]: torch.ones(B) * (-1) for a in attrs}
    w_img = cfg.get("weight_loss_img", 0.0) if ("clip" in models and "dino" in models) else 0.0
    w_face = cfg.get("weight_loss_face", 0.0) if ("face_net" in models and "face_db" in models) else 0.0
    crop = cfg.get("size_aligned_face", 112)
    extra = {k: torch.ones(B) * (-1) for k in ("loss_CLIP", "loss_DINO", "loss_face", "loss")}
    if w_img or w_face:
        with torch.no_grad():
            vb = cfg.get("val_GPU_batch_size", B)
            images_ori = torch.cat([generate_image_no_gradient(tokens, noises[j:j + vb], S, models["eval_text_encoder"], models["eval_unet"],
                                                               vae, sch, gs) for j in range(0, B, vb)])
            ind_o, boxes_o, chips_o = faces(images_ori)
            lo = clf(chips_o[ind_o])
            preds_o, probs_o = [], []
            for name, c0, k in attrs:
                pr = torch.ones(B, k) * (-1)
                pd = torch.ones(B, dtype=torch.long) * (-1)
                p = torch.softmax(lo[:, c0:c0 + k], dim=-1)
                pr[ind_o], pd[ind_o] = p, p.max(dim=-1).indices
                preds_o.append(pd)
                probs_o.append(pr)
            if w_img:
                small_o = resize_small(images_ori, cfg.get("img_size_small", 224))
                clip_o = image_features(models["clip"], small_o, CLIP_IMAGE_MEAN, CLIP_IMAGE_STD)
                dino_o = image_features(models["dino"], small_o, DINO_IMAGE_MEAN, DINO_IMAGE_STD)
            if w_face:
                face_feats_o = get_face_feats(models["face_net"], faces.aligned(images_ori, crop))
    for j in range(N_backward):
        idx = list(range(B))[j * tb:(j + 1) * tb]
        img = generate_image_w_gradient(tokens, noises[idx], S, te, unet, vae, sch, gs)
        ind_j, boxes_j, chips_j = faces(img)
        logits = clf(chips_j[ind_j])
        loss_ij = 0
        for name, c0, k in attrs:
            la = torch.ones(len(idx), k) * (-1)
            la[ind_j] = logits[:, c0:c0 + k]
            t = targets_by_attr[name][idx]
            lf = torch.ones(len(idx)) * (-1)
            w = ((ind_j == True) * (t != -1)).nonzero().view([-1])  # noqa: E712
            lf[w] = F.cross_entropy(la[w], t[w], reduction="none")
            losses[name][idx] = lf.detach()
            loss_ij = loss_ij + lf
        tl = [targets_by_attr[a[0]][idx] for a in attrs]
        if w_img:
            pl = [p[idx] for p in preds_o]
            hooked = apply_grad_hook_face_multi(img, boxes_j, boxes_o[idx], tl, pl, cfg["factors2"])
            small = resize_small(hooked, cfg.get("img_size_small", 224))
            lc = 1 - (image_features(models["clip"], small, CLIP_IMAGE_MEAN, CLIP_IMAGE_STD) * clip_o[idx]).sum(dim=-1)
            ld = 1 - (image_features(models["dino"], small, DINO_IMAGE_MEAN, DINO_IMAGE_STD) * dino_o[idx]).sum(dim=-1)
            dyn = gen_dynamic_weights_multi(ind_j, tl, pl, cfg["factors1"])
            loss_ij = loss_ij + w_img * dyn * (lc + ld)
            extra["loss_CLIP"][idx], extra["loss_DINO"][idx] = lc.detach(), ld.detach()
        if w_face:
            aligned_j = faces.aligned(img, crop)
            sel = (ind_j == True)  # noqa: E712
            for t, p, pr in zip(tl, [p[idx] for p in preds_o], [p[idx] for p in probs_o]):
                sel = sel * (t != -1) * (t == p) * (pr.max(dim=-1).values >= cfg.get("face_conf", 0.8))
            from_ori = sel.nonzero().view([-1]).tolist()
            lface = torch.ones(len(idx)) * (-1)
            if len(from_ori) > 0:
                f1 = get_face_feats(models["face_net"], aligned_j[from_ori])
                lface[from_ori] = 1 - (f1 * face_feats_o[idx][from_ori]).sum(dim=-1)
            from_search = sorted(set((ind_j == True).nonzero().view([-1]).tolist()) - set(from_ori))  # noqa: E712  (exp-3 :2135: every face)
            if len(from_search) > 0:
                f2 = get_face_feats(models["face_net"], aligned_j[from_search])
                lface[from_search] = 1 - (f2 * semantic_search(models["face_db"], f2)).sum(dim=-1)
            loss_ij = loss_ij + w_face * lface
            extra["loss_face"][idx] = lface.detach()
        if torch.is_tensor(loss_ij) and loss_ij.requires_grad:
            loss_ij.mean().backward()
        extra["loss"][idx] = loss_ij.detach() if torch.is_tensor(loss_ij) else extra["loss"][idx]
    return dict(losses=losses, N_backward=N_backward, **extra)
