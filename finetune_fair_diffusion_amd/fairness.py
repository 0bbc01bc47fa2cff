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


# canonical 5-point template of the 112x112 aligned chip (:297-303): eyes, nose, mouth corners
ALIGNED_FACE_LANDMARKS = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366], [41.5493, 92.3655], [70.7299, 92.2041]])


class SyntheticFaceProvider:
    """Deterministic stand-in for the detector side-car: every image has one face whose raw detector
    box is the centred half-size square; the reference's ``expand_bbox(bbox, 0.5, 1)`` then applies
    (:1335).  A real provider returns (indicators [N] bool, boxes [N,4] int, -1 where no face) and, for the
    face-realism term, ``landmarks(images)`` -> [N,5,2] float (x,y) pixel positions (-1 where no face)."""

    def __call__(self, images):
        N, _, H, W = images.shape
        box = expand_bbox([0.25 * W, 0.25 * H, 0.75 * W, 0.75 * H], 0.5, 1)
        return torch.ones(N, dtype=torch.bool), torch.tensor([box] * N, dtype=torch.int32)

    def landmarks(self, images):
        """The canonical template scaled into the raw detector box."""
        N, _, H, W = images.shape
        pts = ALIGNED_FACE_LANDMARKS / 112.0 * np.array([0.5 * W, 0.5 * H]) + np.array([0.25 * W, 0.25 * H])
        return torch.tensor(pts, dtype=torch.float32)[None].repeat(N, 1, 1)


class DetectorFaceProvider:
    """The reference's detector seam behind the provider interface: ``get_face`` (:1192-1215) = insightface's ``FaceAnalysis`` first
    (``get_face_app`` :1306-1353: BGR uint8 image in, largest face by the area of its box clipped to the image, ``expand_bbox(bbox, 0.5, 1)``,
    the detector's 5 key points), and for the images it finds nothing in, face_recognition's CNN detector (``get_face_FR`` :1232-1293:
    ``face_locations(model="cnn", number_of_times_to_upsample=0)`` in (top, right, bottom, left) order, ``expand_bbox(bbox, 1.1, 1)``, five
    points derived from the 68-point "large" landmark model: eye means, last nose-bridge point, top-lip points 0 and 6).

    ``face_app`` / ``face_recognition`` are the two detector objects (``insightface.app.FaceAnalysis`` instance with ``.get``, the
    ``face_recognition`` module); ``from_installed()`` builds them the way the reference does (:936-945; the ONNX execution provider is the CPU one here --
    onnxruntime has no CUDA provider on this platform).  Neither package ships with this
    image: the adaptor is host-side glue, pinned in tests against the reference's own ``get_face`` on scripted detector outputs."""

    def __init__(self, face_app=None, face_recognition=None, fill_value=-1):
        if face_app is None and face_recognition is None:
            raise ValueError("DetectorFaceProvider needs at least one detector (insightface FaceAnalysis and / or the face_recognition module)")
        self.face_app, self.fr, self.fill = face_app, face_recognition, fill_value
        self._seen = []          # results for the last few image batches, by object identity (boxes and landmarks come from one detector pass)

    @classmethod
    def from_installed(cls, det_size=(640, 640), ctx_id=0):
        try:
            from insightface.app import FaceAnalysis
            import face_recognition
        except ImportError as e:
            raise ImportError("--face_provider detector needs the reference's detector side-car packages (insightface, face_recognition); "
                              "neither is part of this build") from e
        app = FaceAnalysis(name="buffalo_l", allowed_modules=["detection"], providers=["CPUExecutionProvider"])
        app.prepare(ctx_id=ctx_id, det_size=det_size)
        return cls(app, face_recognition)

    @staticmethod
    def _largest(boxes_xyxy, dim_max, dim_min=0):
        """get_largest_face_app / get_largest_face_FR (:1217-1230, :1294-1304): first box of maximal clipped area (strict >, start at 0)."""
        if len(boxes_xyxy) == 1:
            return 0
        area_max, idx_max = 0, 0
        for i, b in enumerate(boxes_xyxy):
            area = (min(b[2], dim_max) - max(b[0], dim_min)) * (min(b[3], dim_max) - max(b[1], dim_min))
            if area > area_max:
                area_max, idx_max = area, i
        return idx_max

    def detect(self, images):
        """images [N,3,H,W] in [-1,1] -> (indicators [N] bool, boxes [N,4] int32, landmarks [N,5,2] float32), fill value where no face."""
        for im, res in self._seen:
            if im is images:
                return res
        x = ((images.detach().float() * 0.5 + 0.5) * 255).cpu().permute(0, 2, 3, 1).numpy().astype(np.uint8)
        ind, boxes, lms = [], [], []
        for img in x:
            found = None
            if self.face_app is not None:
                faces = self.face_app.get(img[:, :, [2, 1, 0]])
                if len(faces):
                    f = faces[self._largest([fc["bbox"] for fc in faces], img.shape[0])]
                    found = (expand_bbox(f["bbox"], 0.5, 1), np.asarray(f["kps"], dtype=np.float64))
            if found is None and self.fr is not None:
                locs = self.fr.face_locations(img, model="cnn", number_of_times_to_upsample=0)
                if len(locs):
                    xyxy = [np.array((l[-1],) + tuple(l[:-1])) for l in locs]                  # (top, right, bottom, left) -> (left, top, right, bottom)
                    k = self._largest(xyxy, img.shape[0])
                    lm = self.fr.face_landmarks(img, face_locations=[locs[k]], model="large")[0]
                    pts = np.stack([np.array(lm["left_eye"]).mean(axis=0), np.array(lm["right_eye"]).mean(axis=0), np.array(lm["nose_bridge"][-1]),
                                    np.array(lm["top_lip"][0]), np.array(lm["top_lip"][6])])
                    found = (expand_bbox(xyxy[k], 1.1, 1), pts)
            ind.append(found is not None)
            boxes.append(found[0] if found else [self.fill] * 4)
            lms.append(found[1] if found else np.full((5, 2), float(self.fill)))
        res = (torch.tensor(ind, dtype=torch.bool), torch.tensor(boxes, dtype=torch.int32), torch.tensor(np.stack(lms), dtype=torch.float32))
        self._seen = (self._seen + [(images, res)])[-3:]
        return res

    def __call__(self, images):
        ind, boxes, _ = self.detect(images)
        return ind, boxes

    def landmarks(self, images):
        return self.detect(images)[2]


def umeyama_similarity(src, dst):
    """Least-squares similarity transform (Umeyama 1991) with scale, mapping ``src`` -> ``dst`` points [n,2]; what
    ``skimage.transform.SimilarityTransform().estimate(src, dst)`` stores in ``.params`` (:305-306).  Returns the 3x3 matrix."""
    src, dst = np.asarray(src, dtype=np.float64), np.asarray(dst, dtype=np.float64)
    n, dim = src.shape
    ms, md = src.mean(axis=0), dst.mean(axis=0)
    s0, d0 = src - ms, dst - md
    cov = d0.T @ s0 / n
    sign = np.ones(dim)
    if np.linalg.det(cov) < 0:
        sign[-1] = -1
    U, S, Vt = np.linalg.svd(cov)
    rank = np.linalg.matrix_rank(cov)
    T = np.eye(dim + 1)
    if rank == 0:
        return np.full_like(T, np.nan)
    if rank == dim - 1 and np.linalg.det(U) * np.linalg.det(Vt) <= 0:
        flip = sign.copy()
        flip[-1] = -1
        T[:dim, :dim] = U @ np.diag(flip) @ Vt
    elif rank == dim - 1:
        T[:dim, :dim] = U @ Vt
    else:
        T[:dim, :dim] = U @ np.diag(sign) @ Vt
    scale = (S @ sign) / s0.var(axis=0).sum()
    T[:dim, dim] = md - scale * (T[:dim, :dim] @ ms)
    T[:dim, :dim] *= scale
    return T


def alignment_sampling_matrix(landmarks, H, W, crop=112):
    """image_pipeline (:292-312) as one 2x3 matrix per face: output pixel (x, y, 1) of the ``crop`` x ``crop`` chip -> sampling
    position in input pixels.  Folds together the similarity transform image -> template, kornia 0.7 ``warp_affine``'s
    normalisation of that pixel homography with (size-1), ``affine_grid`` and ``grid_sample`` at ``align_corners=False``:
        base grid  x_n = (2x+1)/crop - 1;  kornia dst pixel  x_d = (x_n+1)(crop-1)/2;  src pixel p = M^-1 x_d;
        normalised g = 2p/(W-1) - 1;  sampled position  ((g+1) W - 1)/2 = p W/(W-1) - 1/2."""
    M = umeyama_similarity(np.asarray(landmarks, dtype=np.float64), ALIGNED_FACE_LANDMARKS * (crop / 112.0))
    Minv = np.linalg.inv(M)
    a = (crop - 1.0) / crop
    to_dst = np.array([[a, 0.0, 0.5 * a], [0.0, a, 0.5 * a], [0.0, 0.0, 1.0]])                 # output pixel index -> kornia dst pixel
    to_samp = np.array([[W / (W - 1.0), 0.0, -0.5], [0.0, H / (H - 1.0), -0.5], [0.0, 0.0, 1.0]])  # src pixel -> grid_sample position
    return (to_samp @ Minv @ to_dst)[:2].reshape(6)


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
        # the reference slices image[:, y0:y1, x0:x1] with (H,W) swapped in the clamps (:1592-1598); a missing original face
        # (box_ori = -1) makes y1/x1 negative, which Python slicing counts from the far edge -- reproduced here explicitly
        x0, y0, x1, y1 = max(bb[0], bo[0], 0), max(bb[1], bo[1], 0), min(bb[2], bo[2], H), min(bb[3], bo[3], W)
        x1 = max(x1 + W, 0) if x1 < 0 else min(x1, W)
        y1 = max(y1 + H, 0) if y1 < 0 else min(y1, H)
        rects.append([x0, y0, x1, y1])
        facs.append(1.0 if (t != -1 and t == p) else factor)
    return torch.tensor(rects, dtype=torch.int32), torch.tensor(facs, dtype=torch.float32)


def _mismatch_factor(targets_i, preds_ori_i, factors):
    """1 when every attribute's target equals the original prediction, else the smallest factor among the mismatching attributes
    (a missing target, -1, never equals a prediction) -- exp-3 :1764-1771 / :1793-1800, exp-4 :1844-1854 / :1882-1892."""
    bad = [f for t, p, f in zip(targets_i, preds_ori_i, factors) if t != p]
    return 1.0 if not bad else float(min(bad))


def gen_dynamic_weights_multi(face_indicators, targets_list, preds_ori_list, factors):
    """exp-3 :1786-1803, exp-4 :1870-1895: images without a face get min(factors) (exp-1's single-attribute version gives them 1)."""
    w = []
    for i, ind in enumerate(face_indicators.tolist()):
        w.append(float(min(factors)) if not ind else _mismatch_factor([t[i].item() for t in targets_list], [p[i].item() for p in preds_ori_list], factors))
    return torch.tensor(w, dtype=torch.float32)


def face_grad_factors_multi(boxes, boxes_ori, targets_list, preds_ori_list, factors, H, W):
    """Multi-attribute ``apply_grad_hook_face`` (exp-3 :1751-1783, exp-4 :1823-1867): same rectangle as exp-1, factor by _mismatch_factor."""
    rects, _ = face_grad_factors(boxes, boxes_ori, targets_list[0], preds_ori_list[0], 1.0, H, W)
    facs = []
    for i, bb in enumerate(boxes.tolist()):
        facs.append(1.0 if all(v == -1 for v in bb) else
                    _mismatch_factor([t[i].item() for t in targets_list], [p[i].item() for p in preds_ori_list], factors))
    return rects, torch.tensor(facs, dtype=torch.float32)


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
    dl = torch.zeros(n, lg.shape[1])
    sel = (face_indicators.cpu() & (targets.cpu() != -1)).nonzero().view(-1)
    if len(sel):
        lp = torch.log_softmax(lg[sel], dim=-1)
        t = targets.cpu()[sel]
        loss[sel] = -lp.gather(1, t[:, None])[:, 0]
        g = lp.exp()
        g[torch.arange(len(sel)), t] -= 1.0
        dl[sel] = g * weights[sel][:, None]
    return loss, dl


# ------------------------------------------------------------------------------------------ multi-attribute (exp-3/4/5)
def _ot_assign(M, counts):
    """Optimal transport from N unit-mass sources to integer-capacity sinks (sum(counts) == N) -- what the reference
    solves with ``ot.emd(ones(N), counts, M)`` (POT, not installable here).  With unit sources and integer sinks the
    LP has an integral optimum: an assignment problem on the sink-replicated cost matrix."""
    from scipy.optimize import linear_sum_assignment
    N, K = M.shape
    cols = np.repeat(np.arange(K), counts)
    assert len(cols) == N
    r, c = linear_sum_assignment(M[:, cols])
    T = np.zeros((N, K))
    T[r, cols[c]] = 1.0
    return T


def _cells(attr_sizes):
    """Cell index = mixed-radix number over the attributes in order (exp-3: g*4+r `:1510`; exp-4: g*8+r*2+a `exp-4:1522`)."""
    import itertools as it
    return list(it.product(*[range(k) for k in attr_sizes]))


@torch.no_grad()
def mc_transport_problem(probs_list, class_cdfs, num_samples_per_device=100, generator=None, age_asymmetric=False):
    """The inputs of the Monte-Carlo transport solves (exp-3 `:1488-1536`, exp-4 `:1517-1569`): returns (idx [n] bool of faces, cost
    matrix M [N, K] fp64 = distance of the probability vectors to the one-hot corners of each cell `:1514-1531`, counts [S, K] int64 =
    drawn cell capacities per Monte-Carlo sample `:1492-1512`, attribute sizes) -- (idx, None, None, sizes) without faces."""
    n = probs_list[0].shape[0]
    sizes = [p.shape[1] for p in probs_list]
    idx = torch.ones(n, dtype=torch.bool)
    for p in probs_list:
        idx &= (p != -1).all(dim=-1)
    N = int(idx.sum())
    if N == 0:
        return idx, None, None, sizes
    P = [p[idx].float().numpy() for p in probs_list]
    cells = _cells(sizes)
    K = len(cells)
    # Monte-Carlo class draws (`:1492-1512`)
    draws = []
    for cdf in class_cdfs:
        u = torch.rand([num_samples_per_device, N], generator=generator)
        cls = torch.zeros_like(u, dtype=torch.long)
        lo = 0.0
        for c, hi in enumerate(cdf):
            if c > 0:
                cls[(u > lo) & (u <= hi)] = c
            lo = hi
        draws.append(cls.numpy())
    radix = np.array([int(np.prod(sizes[a + 1:])) for a in range(len(sizes))])
    # cost matrix: distance of the probability vectors to the one-hot corners of each cell (`:1514-1531`)
    M = np.zeros((N, K))
    for j, cell in enumerate(cells):
        sq = np.zeros(N)
        for a, c in enumerate(cell):
            onehot = np.zeros(sizes[a]); onehot[c] = 1.0
            d = P[a] - onehot
            if age_asymmetric and a == len(sizes) - 1 and c == 1:
                d = d.copy(); d[:, 0] *= 2.0
            sq += (d ** 2).sum(axis=1)
        M[:, j] = np.sqrt(sq)
    counts = np.zeros((num_samples_per_device, K), dtype=np.int64)
    for s in range(num_samples_per_device):
        cell_idx = sum(draws[a][s] * radix[a] for a in range(len(sizes)))
        counts[s] = np.bincount(cell_idx, minlength=K)
    return idx, M, counts, sizes


@torch.no_grad()
def mc_transport_plan(probs_list, class_cdfs, num_samples_per_device=100, generator=None, age_asymmetric=False, device=None):
    """The rank-local half of the multi-attribute dynamic targets (exp-3 `:1488-1536`, exp-4 `:1517-1569`): for each of
    ``num_samples_per_device`` Monte-Carlo draws of a balanced class assignment the faces are optimally transported onto the drawn
    cell counts; returns (idx [n] bool of faces, summed plans [N, K] fp32, attribute sizes) -- or (idx, None, sizes) without faces.
    It depends only on the gathered probabilities, so the step runs it on a worker thread underneath the R2 rollout.

    ``device``: solve the draws on that GPU (``fd_ot_assign_sum``: one wave per draw, exact shortest-augmenting-path assignment in fp64;
    the summed plan stays on the device for the all-reduce); None: the host solver (scipy's assignment -- what the CPU tests and the
    ``FD_OT_HOST`` measurement switch use).  Both are exact: the plans are equal wherever the optimum is unique."""
    idx, M, counts, sizes = mc_transport_problem(probs_list, class_cdfs, num_samples_per_device, generator, age_asymmetric)
    if M is None:
        return idx, None, sizes
    if device is not None:
        from . import ops
        tp = ops.ot_assign_sum(torch.from_numpy(M).to(device), torch.from_numpy(counts.astype(np.int32)).to(device))
        return idx, tp, sizes
    tp = np.zeros(M.shape)
    for s in range(counts.shape[0]):
        tp += _ot_assign(M, counts[s])
    return idx, torch.tensor(tp, dtype=torch.float32), sizes


@torch.no_grad()
def targets_from_plan(idx, tp, sizes):
    """Normalise the (all-reduced) summed plan, marginalise per attribute, argmax / 1 - max (`:1538-1553`)."""
    n = idx.shape[0]
    out_t = [torch.full([n], -1, dtype=torch.long) for _ in sizes]
    out_u = [torch.full([n], -1.0, dtype=torch.float32) for _ in sizes]
    if tp is None:
        return list(zip(out_t, out_u))
    cells = _cells(sizes)
    tp = tp.cpu()
    tp = tp / tp[0, :].sum()
    for a in range(len(sizes)):
        marg = torch.zeros(tp.shape[0], sizes[a])
        for j, cell in enumerate(cells):
            marg[:, cell[a]] += tp[:, j]
        out_t[a][idx] = marg.argmax(dim=-1)
        out_u[a][idx] = 1 - marg.max(dim=-1).values
    return list(zip(out_t, out_u))


@torch.no_grad()
def generate_dynamic_targets_multi(probs_list, class_cdfs, num_samples_per_device=100, generator=None, allreduce=None,
                                   age_asymmetric=False, return_plan=False, device=None):
    """Dynamic targets for several attributes at once (exp-3-debias-gender-race/1-main-debias.py:1459-1569,
    exp-4-debias-gender-race-age/1-main-debias.py:1477-1615).

    probs_list: per attribute a CPU tensor [n, k_a] (-1 rows = no face).  class_cdfs: per attribute the upper CDF
    edges used to turn a uniform draw into a class (gender [0.5, 1], race [.25,.5,.75,1], age [.75, 1]).
    The Monte-Carlo plans (``mc_transport_plan``) are summed (and all-reduced over ranks, ``allreduce`` callable),
    normalised, marginalised per attribute (``targets_from_plan``).  Returns [(targets [n] long, uncertainty [n])] per attribute
    (plus the normalised plan with ``return_plan``).  ``device``: run the transport solves on that GPU (see ``mc_transport_plan``).
    ``age_asymmetric``: exp-4's cost doubles the first age coordinate when the target is the second class (`exp-4:1551-1556`)."""
    idx, tp, sizes = mc_transport_plan(probs_list, class_cdfs, num_samples_per_device, generator, age_asymmetric, device=device)
    if tp is not None and allreduce is not None:
        tp = allreduce(tp)
    res = targets_from_plan(idx, tp, sizes)
    if return_plan:
        return res, (None if tp is None else tp.cpu() / tp[0, :].sum().cpu())
    return res


# per experiment: (factor1 flags, factor2 flags, confidence-level flag) of the regulariser terms, in attribute order
EXPERIMENT_REG_FLAGS = {
    "exp-1": (["factor1"], ["factor2"], "face_gender_confidence_level"),
    "exp-2": (["factor1"], ["factor2"], "face_gender_confidence_level"),
    "exp-3": (["factor1_gender", "factor1_race"], ["factor2_gender", "factor2_race"], "face_gender_race_confidence_level"),
    "exp-4": (["factor1_gender", "factor1_race", "factor1_age"], ["factor2_gender", "factor2_race", "factor2_age"], "face_gender_race_age_confidence_level"),
    "exp-5": (["factor1_gender", "factor1_race"], ["factor2_gender", "factor2_race"], "face_gender_race_confidence_level"),
}

EXPERIMENT_ATTRS = {
    # name: (classifier logits, [(attribute, first logit column, width)], class CDF edges, exp-4 age asymmetry)
    "exp-1": (80, [("gender", 40, 2)], None, False),
    "exp-2": (80, [("gender", 40, 2)], None, False),      # same classifier and binary targets as exp-1; what is trained differs (prefix.py)
    "exp-3": (6, [("gender", 0, 2), ("race", 2, 4)], [[0.5, 1.0], [0.25, 0.5, 0.75, 1.0]], False),
    "exp-4": (8, [("gender", 0, 2), ("race", 2, 4), ("age", 6, 2)], [[0.5, 1.0], [0.25, 0.5, 0.75, 1.0], [0.75, 1.0]], True),
    "exp-5": (6, [("gender", 0, 2), ("race", 2, 4)], [[0.5, 1.0], [0.25, 0.5, 0.75, 1.0]], False),
}
