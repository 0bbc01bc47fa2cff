"""Host-side fairness logic of the reference's training step, restated (not imported) from
exp-1-debias-gender/1-main-debias.py -- the pieces that are NOT arithmetic on big tensors:
face-box expansion (:238-265), the face-provider seam that replaces the insightface/dlib detectors
(:1192-1353), dynamic targets from the batch's class probabilities (:1403-1447),
per-image loss weights (:1619-1633) and the distributional-alignment loss gradient (:1912-1933).
Pinned against the reference's own outputs by tests/golden/reference_pure_functions.json.
"""
import math

import numpy as np
import scipy.stats
import torch


def expand_bbox(bbox, expand_coef, target_ratio):
    """:238-265 -- [x0,y0,x1,y1] -> expanded integer box with height/width == target_ratio."""
    bw, bh = bbox[2] - bbox[0], bbox[3] - bbox[1]
    if bh / bw > target_ratio:
        more_h = bh * expand_coef
        more_w = (bh + more_h) / target_ratio - bw
    else:
        more_w = bw * expand_coef
        more_h = (bw + more_w) * target_ratio - bh
    return [int(round(bbox[0] - more_w * 0.5)), int(round(bbox[1] - more_h * 0.5)),
            int(round(bbox[2] + more_w * 0.5)), int(round(bbox[3] + more_h * 0.5))]


class SyntheticFaceProvider:
    """Deterministic stand-in for the detector side-car: every image has one face whose raw detector
    box is the centred half-size square; the reference's ``expand_bbox(bbox, 0.5, 1)`` then applies
    (:1335).  A real provider returns (indicators [N] bool, boxes [N,4] int, -1 where no face)."""

    def __call__(self, images):
        N, _, H, W = images.shape
        box = expand_bbox([0.25 * W, 0.25 * H, 0.75 * W, 0.75 * H], 0.5, 1)
        return torch.ones(N, dtype=torch.bool), torch.tensor([box] * N, dtype=torch.int32)


@torch.no_grad()
def generate_dynamic_targets(probs, target_ratio=0.5, w_uncertainty=False):
    """:1403-1447 on a CPU tensor ``probs`` [n,2] (-1 rows = no face)."""
    probs = probs.detach().float().cpu()
    idxs = (probs != -1).all(dim=-1)
    p = probs[idxs]
    rank = torch.argsort(torch.argsort(p[:, 1]))
    targets = (rank >= (rank.shape[0] * target_ratio)).long()
    targets_all = torch.full([probs.shape[0]], -1, dtype=torch.long)
    targets_all[idxs] = targets
    if not w_uncertainty:
        return targets_all
    unc = torch.full([p.shape[0]], -1.0, dtype=probs.dtype)
    n = p.shape[0]
    unc[targets == 1] = torch.tensor(1 - scipy.stats.binom.cdf(rank[targets == 1].numpy(), n, 1 - target_ratio)).to(probs.dtype)
    unc[targets == 0] = torch.tensor(scipy.stats.binom.cdf(rank[targets == 0].numpy(), n, target_ratio)).to(probs.dtype)
    unc_all = torch.full([probs.shape[0]], -1.0, dtype=probs.dtype)
    unc_all[idxs] = unc
    return targets_all, unc_all


def gen_dynamic_weights(face_indicators, targets, preds_ori, factor=0.2):
    """:1619-1633."""
    w = []
    for ind, t, p in zip(face_indicators.tolist(), targets.tolist(), preds_ori.tolist()):
        w.append(1.0 if (not ind) or (t != -1 and t == p) else factor)
    return torch.tensor(w, dtype=torch.float32)


def face_grad_factors(boxes, boxes_ori, targets, preds_ori, factor, H, W):
    """Backward-only effect of ``apply_grad_hook_face`` (:1584-1617): for every image the rectangle
    ``box ∩ box_ori`` (clipped) and the factor its image-gradient is multiplied by.  Returns
    (rects [N,4] int (x0,y0,x1,y1; empty if no face), factors [N])."""
    rects, facs = [], []
    for bb, bo, t, p in zip(boxes.tolist(), boxes_ori.tolist(), targets.tolist(), preds_ori.tolist()):
        if all(v == -1 for v in bb):
            rects.append([0, 0, 0, 0]); facs.append(1.0)
            continue
        rects.append([max(bb[0], bo[0], 0), max(bb[1], bo[1], 0), min(bb[2], bo[2], H), min(bb[3], bo[3], W)])
        facs.append(1.0 if (t != -1 and t == p) else factor)
    return torch.tensor(rects, dtype=torch.int32), torch.tensor(facs, dtype=torch.float32)


def microbatch_weights(B, train_GPU_batch_size):
    """The reference back-propagates ``loss_ij.mean()`` per micro-batch j (:1889-1933) and later divides
    the gradient by N_backward (:2005): image i of a chunk of n_j images carries weight 1/n_j.
    Returns (weights [B], N_backward)."""
    nb = math.ceil(B / train_GPU_batch_size)
    w = torch.empty(B, dtype=torch.float32)
    for j in range(nb):
        idx = list(range(B))[j * train_GPU_batch_size:(j + 1) * train_GPU_batch_size]
        w[idx] = 1.0 / len(idx)
    return w, nb


def fair_loss_and_grad(logits_attr, targets, face_indicators, weights):
    """Distributional-alignment loss (:1912-1915, CE with reduction='none' scattered into a -1-filled
    vector) and its gradient w.r.t. the attribute logits under the per-image weights.
    logits_attr [n,2] (any float dtype, any device), targets [n] long (-1 = no target).
    Returns (loss_fair [n] fp32 with -1 sentinels, dlogits [n,2] fp32)."""
    lg = logits_attr.detach().float().cpu()
    n = lg.shape[0]
    loss = torch.full([n], -1.0)
    dl = torch.zeros(n, 2)
    sel = (face_indicators.cpu() & (targets.cpu() != -1)).nonzero().view(-1)
    if len(sel):
        lp = torch.log_softmax(lg[sel], dim=-1)
        t = targets.cpu()[sel]
        loss[sel] = -lp.gather(1, t[:, None])[:, 0]
        g = lp.exp()
        g[torch.arange(len(sel)), t] -= 1.0
        dl[sel] = g * weights[sel][:, None]
    return loss, dl
