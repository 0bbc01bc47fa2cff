"""Device-resident restatement of the exp-1 (single binary attribute) loss assembly of fairness.py -- dynamic targets (:1403-1447), the
distributional-alignment loss and its logit gradient (:1912-1915), the dynamic weights (:1619-1633), the gradient-hook factors (:1584-1617) and
the face-realism bookkeeping (:1917-1932) -- as torch operations on TINY tensors (at most world*B rows) that never leave the device, so the
step's tail has no host read-back between the two rollouts and the end of the step.  fairness.py stays the host statement
(pinned by the reference-executed goldens); tests/test_cpu.py holds every function here equal to it on CPU tensors.  Tie handling: the
reference ranks with an unstable argsort whose tie order is implementation-defined; here ties rank by image index (stable sort)."""
import numpy as np
import scipy.stats
import torch


def binomial_tables(n_max, target_ratio=0.5):
    """(cdf0 [n_max+1, n_max], cdf1 [n_max+1, n_max]) fp32: the uncertainty of a target-0 / target-1 image of rank r among n faces,
    ``binom.cdf(r, n, ratio)`` and ``1 - binom.cdf(r, n, 1 - ratio)`` (:1433-1442), evaluated in fp64 by scipy exactly as the host code does."""
    c0 = np.zeros((n_max + 1, max(n_max, 1)), dtype=np.float64)
    c1 = np.zeros_like(c0)
    r = np.arange(max(n_max, 1))
    for n in range(n_max + 1):
        c0[n] = scipy.stats.binom.cdf(r, n, target_ratio)
        c1[n] = 1 - scipy.stats.binom.cdf(r, n, 1 - target_ratio)
    return torch.tensor(c0).to(torch.float32), torch.tensor(c1).to(torch.float32)


def dynamic_targets(probs, tables, threshold=None, target_ratio=0.5):
    """``generate_dynamic_targets(probs, w_uncertainty=True)`` for probs [n,2] on any device (-1 rows = no face); ``tables`` from
    ``binomial_tables(n)`` on the same device.  Returns (targets [n] long, -1 where no face -- and, with ``threshold``, where the uncertainty
    exceeds it (:1837) --, uncertainty [n] fp32, -1 where no face)."""
    n = probs.shape[0]
    valid = (probs != -1).all(dim=-1)
    key = torch.where(valid, probs[:, 1].float(), torch.full_like(probs[:, 1], float("inf"), dtype=torch.float32))
    order = torch.argsort(key, stable=True)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(n, device=probs.device)
    nv = valid.sum()
    t = (rank.to(torch.float32) >= nv.to(torch.float32) * target_ratio).long()
    r = rank.clamp(max=tables[0].shape[1] - 1)
    unc = torch.where(t == 1, tables[1][nv, r], tables[0][nv, r])
    targets = torch.where(valid, t, torch.full_like(t, -1))
    unc = torch.where(valid, unc, torch.full_like(unc, -1.0))
    if threshold is not None:
        targets = torch.where(unc > threshold, torch.full_like(targets, -1), targets)
    return targets, unc


def fair_loss_and_grad(logits, targets, face, weights):
    """``fairness.fair_loss_and_grad`` on device tensors: logits [n,k] fp32 (any values where no face), targets [n] long, face [n] bool,
    weights [n] fp32 -> (loss [n] with -1 sentinels, dlogits [n,k])."""
    sel = face & (targets != -1)
    lp = torch.log_softmax(logits.float(), dim=-1)
    tt = targets.clamp(min=0)
    loss = torch.where(sel, -lp.gather(1, tt[:, None])[:, 0], torch.full_like(lp[:, 0], -1.0))
    g = lp.exp() - torch.nn.functional.one_hot(tt, logits.shape[1]).to(lp.dtype)
    dl = torch.where(sel[:, None], g * weights[:, None], torch.zeros_like(g))
    return loss, dl


def dynamic_weights(face, targets, preds_ori, factor):
    """``gen_dynamic_weights`` (:1619-1633): 1 where there is no face or the target equals the original prediction, else ``factor``."""
    keep = (~face) | ((targets != -1) & (targets == preds_ori))
    return torch.where(keep, torch.ones_like(targets, dtype=torch.float32), torch.full_like(targets, factor, dtype=torch.float32))


def hook_factors(has_box, targets, preds_ori, factor):
    """The factor half of ``face_grad_factors`` (the rectangles depend on the boxes only and stay on the host): 1 where the image has no box or
    the target equals the original prediction, else ``factor``."""
    keep = (~has_box) | ((targets != -1) & (targets == preds_ori))
    return torch.where(keep, torch.ones_like(targets, dtype=torch.float32), torch.full_like(targets, factor, dtype=torch.float32))


def probs_preds(logits_sel, sel, n, c0, k):
    """Scatter of get_face_gender (:1369-1401): logits of the images with a face (rows ``sel``) -> (probs [n,k] -1-filled, preds [n] -1-filled,
    logits [n,k] -1-filled), all on the logits' device."""
    dev = logits_sel.device
    la = logits_sel[:, c0:c0 + k].float()
    p = torch.softmax(la, dim=-1)
    probs = torch.full((n, k), -1.0, dtype=torch.float32, device=dev)
    preds = torch.full((n,), -1, dtype=torch.long, device=dev)
    lg = torch.full((n, k), -1.0, dtype=torch.float32, device=dev)
    probs[sel] = p
    preds[sel] = p.max(dim=-1).indices
    lg[sel] = la
    return probs, preds, lg
