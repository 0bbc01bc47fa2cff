"""Generate golden vectors by EXECUTING the reference's own pure functions.

Run in the build container only (needs /root/reference); the outputs
(tests/golden/reference_pure_functions.json, reference_cli.json) are committed and
are what travels to the GPU box.  Functions nested inside ``main`` are lifted by AST
(source segment -> exec) exactly as SURVEY.md section 8c describes; module-level
functions are lifted the same way so no third-party import of the reference runs.
No reference source text is stored: only inputs and outputs.
"""
import argparse
import ast
import itertools
import json
import math
import os
import sys

import numpy as np
import scipy
import scipy.stats
import torch
import yaml

REF = "/root/reference/exp-1-debias-gender/1-main-debias.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def lift(names, ref=None):
    src = open(ref or REF).read()
    tree = ast.parse(src)
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            node.decorator_list = []
            found[node.name] = ast.get_source_segment(src, node)
    ns = dict(torch=torch, np=np, scipy=scipy, itertools=itertools, math=math, argparse=argparse, yaml=yaml, os=os)
    for n in [x for x in ["make_grad_hook"] if x in names] + [x for x in names if x != "make_grad_hook"]:
        seg = found[n]
        seg = "\n".join(l for l in seg.splitlines() if not l.strip().startswith("@"))
        import textwrap
        exec(textwrap.dedent(seg), ns)
    return ns


def main():
    ns = lift(["make_grad_hook", "expand_bbox", "generate_dynamic_targets", "gen_dynamic_weights",
               "apply_grad_hook_face", "get_face_gender", "parse_args"])
    out = {}
    # expand_bbox -------------------------------------------------------------
    rng = np.random.RandomState(0)
    cases = [[100.2, 120.7, 300.9, 380.1], [128.0, 128.0, 384.0, 384.0], [-10.5, 3.2, 200.1, 150.9], [32.0, 32.0, 96.0, 96.0]]
    for _ in range(12):
        x0, y0 = rng.uniform(-20, 300, 2)
        cases.append([float(x0), float(y0), float(x0 + rng.uniform(20, 250)), float(y0 + rng.uniform(20, 250))])
    eb = []
    for c in cases:
        for coef, ratio in [(0.5, 1), (1.1, 1), (0.3, 1.25)]:
            eb.append(dict(bbox=c, expand_coef=coef, target_ratio=ratio, out=ns["expand_bbox"](np.array(c), coef, ratio)))
    out["expand_bbox"] = eb
    # generate_dynamic_targets --------------------------------------------------
    gdt = []
    for seed, n, nmiss in [(0, 8, 1), (1, 8, 0), (2, 16, 3), (3, 24, 2), (4, 5, 0), (5, 64, 5), (6, 1, 0), (7, 12, 12)]:
        g = torch.Generator().manual_seed(seed)
        p1 = torch.rand(n, generator=g)
        probs = torch.stack([1 - p1, p1], dim=-1)
        miss = torch.randperm(n, generator=g)[:nmiss]
        probs[miss] = -1
        t, u = ns["generate_dynamic_targets"](probs, target_ratio=0.5, w_uncertainty=True)
        t2 = ns["generate_dynamic_targets"](probs, target_ratio=0.5, w_uncertainty=False)
        assert torch.equal(t, t2)
        gdt.append(dict(probs=probs.tolist(), targets=t.tolist(), uncertainty=u.tolist()))
    out["generate_dynamic_targets"] = gdt
    # gen_dynamic_weights -------------------------------------------------------
    gdw = []
    for seed in range(4):
        g = torch.Generator().manual_seed(100 + seed)
        n = 8
        ind = torch.rand(n, generator=g) > 0.2
        targets = torch.randint(-1, 2, (n,), generator=g)
        preds = torch.randint(0, 2, (n,), generator=g)
        probs = torch.rand(n, 2, generator=g)
        w = ns["gen_dynamic_weights"](ind, targets, preds, probs, factor=0.2)
        gdw.append(dict(face_indicators=ind.tolist(), targets=targets.tolist(), preds_ori=preds.tolist(), weights=w.tolist()))
    out["gen_dynamic_weights"] = gdw
    # apply_grad_hook_face: forward identity + gradient mask -----------------------
    agh = []
    for seed in range(4):
        g = torch.Generator().manual_seed(200 + seed)
        n, H = 4, 32
        images = torch.randn(n, 3, H, H, generator=g, requires_grad=True)
        bb = torch.tensor([[4, 6, 20, 24], [-1, -1, -1, -1], [0, 0, 40, 40], [10, 3, 30, 17]])
        bbo = torch.tensor([[8, 2, 28, 22], [3, 3, 9, 9], [5, 5, 25, 25], [-3, -2, 12, 40]])
        if seed == 3:   # faces that the original model's image does not have: bbox_ori = -1 turns into Python's negative slice ends
            bb = torch.tensor([[4, 6, 20, 24], [2, 2, 30, 30], [0, 0, 40, 40], [10, 3, 30, 17]])
            bbo = torch.tensor([[-1, -1, -1, -1], [-1, -1, -1, -1], [5, 5, 25, 25], [-1, -1, -1, -1]])
        targets = torch.tensor([1, 0, -1, 0])
        preds = torch.tensor([1, 0, 1, 1])
        probs = torch.rand(n, 2, generator=g)
        y = ns["apply_grad_hook_face"](images, bb, bbo, targets, preds, probs, factor=0.2)
        gw = torch.randn(y.shape, generator=g)
        (y * gw).sum().backward()
        ratio = (images.grad / gw)
        agh.append(dict(seed=200 + seed, bbox=bb.tolist(), bbox_ori=bbo.tolist(), targets=targets.tolist(), preds_ori=preds.tolist(),
                        max_abs_fwd_diff=float((y - images).abs().max()),
                        grad_ratio_ch0=ratio[:, 0].round(decimals=4).tolist()))
    out["apply_grad_hook_face"] = agh
    # get_face_gender scatter with a linear stand-in classifier --------------------
    gfg = []
    for seed in range(3):
        g = torch.Generator().manual_seed(300 + seed)
        n = 6
        W = torch.randn(80, 12, generator=g)
        ns["gender_classifier"] = lambda x, W=W: x.flatten(1) @ W.t()
        chips = torch.randn(n, 3, 2, 2, generator=g)
        sel = torch.tensor([True, False, True, True, False, True]) if seed else torch.zeros(n, dtype=torch.bool)
        preds, probs, logits = ns["get_face_gender"](chips, selector=sel, fill_value=-1)
        gfg.append(dict(W=W.tolist(), chips=chips.tolist(), selector=sel.tolist(), preds=preds.tolist(), probs=probs.tolist(), logits=logits.tolist()))
    out["get_face_gender"] = gfg
    multi_attribute_goldens(out)
    json.dump(out, open(os.path.join(HERE, "reference_pure_functions.json"), "w"))

    # CLI: defaults + YAML overlays of every exp-1 config ---------------------------
    cli = {}
    os.environ.pop("LOCAL_RANK", None)
    cli["defaults"] = vars(ns["parse_args"]([]))
    cfg_dir = "/root/reference/exp-1-debias-gender/configs"
    for f in ["debias-unet.yaml", "debias-text-encoder.yaml", "debias-text-encoder-and-unet.yaml"]:
        cli[f] = dict(yaml=yaml.safe_load(open(os.path.join(cfg_dir, f))), args=vars(ns["parse_args"](["--config", os.path.join(cfg_dir, f)])))
        cli[f]["args"]["config"] = f
    json.dump(cli, open(os.path.join(HERE, "reference_cli.json"), "w"), indent=1, sort_keys=True)
    print("wrote golden vectors:", {k: len(v) for k, v in out.items()}, len(cli["defaults"]), "flags")

    # CLI of the multi-attribute experiments (exp-3/4/5): defaults + every YAML config ----------
    multi = {}
    for exp, d in [("exp-2", "exp-2-debias-gender-token"), ("exp-3", "exp-3-debias-gender-race"), ("exp-4", "exp-4-debias-gender-race-age"),
                   ("exp-5", "exp-5-debias-gender-race-multi-concepts")]:
        pa = lift(["parse_args"], ref=f"/root/reference/{d}/1-main-debias.py")["parse_args"]
        e = dict(defaults=vars(pa([])))
        cdir = f"/root/reference/{d}/configs"
        for f in sorted(os.listdir(cdir)):
            if f.endswith(".yaml") and "compute_environment" not in yaml.safe_load(open(os.path.join(cdir, f))):   # skip accelerate launcher configs
                e[f] = dict(yaml=yaml.safe_load(open(os.path.join(cdir, f))), args=vars(pa(["--config", os.path.join(cdir, f)])))
                e[f]["args"]["config"] = f
        multi[exp] = e
    json.dump(multi, open(os.path.join(HERE, "reference_cli_multi.json"), "w"), indent=1, sort_keys=True)
    print("wrote multi-attribute CLI goldens:", {k: len(v["defaults"]) for k, v in multi.items()})

    sfnet_golden()


def multi_attribute_goldens(out):
    """exp-3 / exp-4 versions of gen_dynamic_weights and apply_grad_hook_face (2 and 3 attributes), lifted from their own scripts."""
    for exp, d, na in [("exp3", "exp-3-debias-gender-race", 2), ("exp4", "exp-4-debias-gender-race-age", 3)]:
        ns = lift(["make_grad_hook", "gen_dynamic_weights", "apply_grad_hook_face"], ref=f"/root/reference/{d}/1-main-debias.py")
        ns["itertools"] = itertools
        factors1, factors2 = [0.2, 0.6, 0.5][:na], [0.25, 0.3, 0.1][:na]
        names = ["gender", "race", "age"][:na]
        widths = [2, 4, 2][:na]
        gdw, agh = [], []
        for seed in range(4):
            g = torch.Generator().manual_seed(500 + seed)
            n = 6
            ind = torch.rand(n, generator=g) > 0.25
            T = [torch.randint(-1, w, (n,), generator=g) for w in widths]
            P = [torch.randint(0, w, (n,), generator=g) for w in widths]
            for k in range(na):      # make matches likely
                m = torch.rand(n, generator=g) < 0.5
                T[k][m] = P[k][m]
            probs = [torch.rand(n, w, generator=g) for w in widths]
            args, kw = [ind], {}
            for k in range(na):
                args += [T[k], P[k], probs[k]]
                kw[f"factor_{names[k]}"] = factors1[k]
            w = ns["gen_dynamic_weights"](*args, **kw)
            gdw.append(dict(face_indicators=ind.tolist(), targets=[t.tolist() for t in T], preds_ori=[p.tolist() for p in P], factors=factors1,
                            weights=[round(float(v), 6) for v in w]))
            H = 24
            images = torch.randn(n, 3, H, H, generator=g, requires_grad=True)
            bb = torch.tensor([[4, 6, 20, 22], [-1, -1, -1, -1], [0, 0, 30, 30], [10, 3, 20, 17], [2, 2, 12, 12], [5, 5, 19, 23]])
            bbo = torch.tensor([[8, 2, 28, 20], [3, 3, 9, 9], [5, 5, 25, 25], [-3, -2, 12, 40], [-1, -1, -1, -1], [6, 4, 18, 20]])
            args, kw = [images, bb, bbo], {}
            for k in range(na):
                args += [T[k], P[k], probs[k]]
                kw[f"factor_{names[k]}"] = factors2[k]
            y = ns["apply_grad_hook_face"](*args, **kw)
            gw = torch.randn(y.shape, generator=g)
            (y * gw).sum().backward()
            agh.append(dict(seed=500 + seed, bbox=bb.tolist(), bbox_ori=bbo.tolist(), targets=[t.tolist() for t in T], preds_ori=[p.tolist() for p in P],
                            factors=factors2, grad_ratio_ch0=(images.grad / gw)[:, 0].round(decimals=4).tolist()))
        out[f"gen_dynamic_weights_{exp}"] = gdw
        out[f"apply_grad_hook_face_{exp}"] = agh


def sfnet_golden():
    """opensphere's own ``sfnet20`` (vendored under /root/reference/opensphere, pure torch) on the build's seeded synthetic weights and
    a seeded input: pins oracle/nn_sfnet.py::SFNet20 (same state-dict keys) to the reference's module."""
    sys.path.insert(0, "/root/reference")
    sys.path.insert(0, "/root/reference/opensphere")
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from opensphere.model.backbone.sfnet import sfnet20
    from finetune_fair_diffusion_amd import weights as W
    sd = W.synthetic_state_dict(W.sfnet20_param_shapes(), seed=31)
    net = sfnet20().eval()
    net.load_state_dict(sd, strict=True)
    x = torch.rand(2, 3, 112, 112, generator=torch.Generator().manual_seed(77)) * 2 - 1
    with torch.no_grad():
        y = net(x)
        y2 = net(torch.flip(x, [3]))
    json.dump(dict(weights_seed=31, input_seed=77, n_params=sum(p.numel() for p in net.parameters()), keys=list(net.state_dict().keys()),
                   out=y.tolist(), out_flipped=y2.tolist()), open(os.path.join(HERE, "reference_sfnet20.json"), "w"))
    print("wrote sfnet20 golden:", tuple(y.shape), float(y.abs().max()))


def face_provider_golden():
    """The detector seam of the training step, ``get_face`` (:1192-1215) = ``get_face_app`` (insightface, :1306-1353) with ``get_face_FR``
    (face_recognition's CNN detector, :1232-1293) for the images the first one misses, run on SCRIPTED detector outputs: the two detector
    packages are test doubles that replay the detections listed in the fixture (neither package nor its weights exist here); everything
    between detector output and the tensors the step consumes -- largest-face choice, box order, expand_bbox coefficients, the five
    landmarks taken from the 68-point set, fill values -- is the reference's own code.  ``image_pipeline`` / ``crop_face`` outputs are not
    part of this fixture (pinned elsewhere: test_face_alignment_*, crop goldens)."""
    import types
    ns = lift(["expand_bbox", "get_largest_face_app", "get_largest_face_FR", "get_face_app", "get_face_FR", "get_face"])
    rng = np.random.default_rng(2024)
    H = W = 96

    def rand_box(big=False):
        x0, y0 = rng.uniform(-6, 50, 2)
        w, h = rng.uniform(20, 60 if big else 34, 2)
        return [float(x0), float(y0), float(x0 + w), float(y0 + h)]

    def rand_pts(n):
        return [[float(a), float(b)] for a, b in rng.uniform(10, 86, (n, 2))]

    cases = []
    for c in range(6):
        N = 5
        app, fr = [], []
        for i in range(N):
            k = [1, 0, 3, 0, 2][(i + c) % 5]            # number of insightface detections for image i
            app.append([dict(bbox=rand_box(j == 1), kps=rand_pts(5)) for j in range(k)])
        for i in range(N):
            k = [2, 0, 1, 3, 0][(i + 2 * c) % 5]        # face_recognition detections (asked only where insightface found none)
            locs = []
            for j in range(k):
                b = rand_box(j == 0)
                locs.append([int(b[1]), int(b[2]), int(b[3]), int(b[0])])            # (top, right, bottom, left)
            lms = [dict(left_eye=rand_pts(6), right_eye=rand_pts(6), nose_bridge=rand_pts(4), top_lip=rand_pts(12)) for _ in locs]
            fr.append(dict(locations=locs, landmarks=lms))
        cases.append(dict(H=H, W=W, app=app, fr=fr))

    out = []
    for case in cases:
        images = torch.zeros(len(case["app"]), 3, case["H"], case["W"])
        app_q, fr_q = list(case["app"]), [f for f, a in zip(case["fr"], case["app"]) if len(a) == 0]
        state = dict(fr_cur=None)

        def app_get(img_bgr):
            return [dict(bbox=np.array(d["bbox"]), kps=np.array(d["kps"])) for d in app_q.pop(0)]

        def fr_locations(img, model, number_of_times_to_upsample):
            assert model == "cnn" and number_of_times_to_upsample == 0
            state["fr_cur"] = fr_q.pop(0)
            return [tuple(l) for l in state["fr_cur"]["locations"]]

        def fr_landmarks(img, face_locations, model):
            assert model == "large" and len(face_locations) == 1
            i = [tuple(l) for l in state["fr_cur"]["locations"]].index(tuple(face_locations[0]))
            return [state["fr_cur"]["landmarks"][i]]
        ns["face_app"] = types.SimpleNamespace(get=app_get)
        ns["face_recognition"] = types.SimpleNamespace(face_locations=fr_locations, face_landmarks=fr_landmarks)
        ns["args"] = types.SimpleNamespace(size_face=8, size_aligned_face=6)
        ns["image_pipeline"] = lambda img, lm: torch.zeros(3, 6, 6)               # chips are not part of this fixture (torchvision / kornia /
        ns["crop_face"] = lambda img, bbox, target_size, fill_value: torch.zeros(3, 8, 8)   # skimage are absent here)
        ind, boxes, _, lms, _ = ns["get_face"](images, fill_value=-1)
        out.append(dict(case, indicators=ind.tolist(), boxes=[[int(v) for v in b] for b in boxes.tolist()], landmarks=lms.tolist()))
    json.dump(out, open(os.path.join(HERE, "reference_face_provider.json"), "w"))
    print("wrote face-provider golden:", len(out), "cases;", sum(sum(c["indicators"]) for c in out), "faces of", sum(len(c["indicators"]) for c in out))


if __name__ == "__main__":
    main()
    face_provider_golden()
