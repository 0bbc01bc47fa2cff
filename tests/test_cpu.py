"""CPU suite (no GPU): the oracle and the product's host logic against the golden vectors produced by the
reference's own functions, parameter inventories against the oracle modules, the C-ABI library's exported
symbols against the public header, scheduler analytics, and the CLI surface."""
import ctypes
import json
import math
import os
import re
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_pure_functions.json")))
GOLD_CLI = json.load(open(os.path.join(HERE, "golden", "reference_cli.json")))


# ------------------------------------------------------------------ golden vectors (reference's own outputs)
@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_expand_bbox_golden(impl):
    if impl == "oracle":
        from oracle.fair_step import expand_bbox
    else:
        from finetune_fair_diffusion_amd.fairness import expand_bbox
    for c in GOLD["expand_bbox"]:
        assert expand_bbox(c["bbox"], c["expand_coef"], c["target_ratio"]) == c["out"]


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_dynamic_targets_golden(impl):
    if impl == "oracle":
        from oracle.fair_step import generate_dynamic_targets
    else:
        from finetune_fair_diffusion_amd.fairness import generate_dynamic_targets
    for c in GOLD["generate_dynamic_targets"]:
        probs = torch.tensor(c["probs"]).reshape(-1, 2)
        t, u = generate_dynamic_targets(probs, target_ratio=0.5, w_uncertainty=True)
        assert t.tolist() == c["targets"]
        assert np.allclose(u.numpy(), np.array(c["uncertainty"]), atol=1e-6)


def test_dynamic_weights_golden():
    from finetune_fair_diffusion_amd.fairness import gen_dynamic_weights as prod
    from oracle.fair_step import gen_dynamic_weights as orc
    for c in GOLD["gen_dynamic_weights"]:
        args = (torch.tensor(c["face_indicators"]), torch.tensor(c["targets"]), torch.tensor(c["preds_ori"]))
        assert np.allclose(prod(*args, factor=0.2).numpy(), c["weights"])
        assert np.allclose(orc(*args, factor=0.2).numpy(), c["weights"])


def test_grad_hook_face_golden():
    """Oracle's hook (autograd) and the product's rectangle/factor form both reproduce the reference's gradient mask."""
    from finetune_fair_diffusion_amd.fairness import face_grad_factors
    from oracle.fair_step import apply_grad_hook_face
    for c in GOLD["apply_grad_hook_face"]:
        g = torch.Generator().manual_seed(c["seed"])
        images = torch.randn(4, 3, 32, 32, generator=g, requires_grad=True)
        bb, bbo = torch.tensor(c["bbox"]), torch.tensor(c["bbox_ori"])
        t, p = torch.tensor(c["targets"]), torch.tensor(c["preds_ori"])
        y = apply_grad_hook_face(images, bb, bbo, t, p, factor=0.2)
        assert float((y - images).abs().max()) == 0.0 == c["max_abs_fwd_diff"]
        gw = torch.ones_like(y)
        (y * gw).sum().backward()
        ref = torch.tensor(c["grad_ratio_ch0"])
        assert torch.allclose(images.grad[:, 0], ref, atol=1e-4)
        rects, facs = face_grad_factors(bb, bbo, t, p, 0.2, 32, 32)
        mask = torch.ones(4, 32, 32)
        for i, (r, f) in enumerate(zip(rects.tolist(), facs.tolist())):
            mask[i, r[1]:r[3], r[0]:r[2]] = f
        assert torch.allclose(mask, ref, atol=1e-4)


def test_face_gender_scatter_golden():
    from oracle.fair_step import get_face_gender
    for c in GOLD["get_face_gender"]:
        W = torch.tensor(c["W"])
        clf = lambda x, W=W: x.flatten(1) @ W.t()  # noqa: E731
        chips, sel = torch.tensor(c["chips"]), torch.tensor(c["selector"])
        preds, probs, logits = get_face_gender(clf, chips, selector=sel, fill_value=-1)
        assert preds.tolist() == c["preds"]
        assert np.allclose(probs.numpy(), np.array(c["probs"]).reshape(probs.shape), atol=1e-6)
        assert np.allclose(logits.numpy(), np.array(c["logits"]).reshape(logits.shape), atol=1e-5)


def test_fair_loss_matches_cross_entropy():
    from finetune_fair_diffusion_amd.fairness import fair_loss_and_grad, microbatch_weights
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(8, 2, generator=g, requires_grad=True)
    targets = torch.tensor([0, 1, -1, 1, 0, -1, 1, 0])
    ind = torch.tensor([True, True, True, False, True, True, True, True])
    w, nb = microbatch_weights(8, 3)
    assert nb == 3 and np.allclose(w.numpy(), [1 / 3] * 6 + [1 / 2] * 2)
    loss, dl = fair_loss_and_grad(logits, targets, ind, w)
    # reference arithmetic: per chunk, CE scattered into a -1 vector, mean over the chunk, backward (:1912-1933)
    total = 0
    for j in range(3):
        idx = list(range(8))[3 * j:3 * j + 3]
        lf = torch.ones(len(idx)) * -1
        sel = [k for k, i in enumerate(idx) if ind[i] and targets[i] != -1]
        if sel:
            ii = torch.tensor([idx[k] for k in sel])
            lf = lf.index_put((torch.tensor(sel),), torch.nn.functional.cross_entropy(logits[ii], targets[ii], reduction="none"))
        total = total + lf.mean()
        assert torch.allclose(loss[idx], lf.detach(), atol=1e-6)
    total.backward()
    assert torch.allclose(dl, logits.grad, atol=1e-6)


# ------------------------------------------------------------------ CLI surface
def test_cli_defaults_and_yaml_overlays(tmp_path):
    import yaml
    from finetune_fair_diffusion_amd.cli import parse_args
    os.environ.pop("LOCAL_RANK", None)
    d = vars(parse_args([]))
    assert d == GOLD_CLI["defaults"]
    for f in ["debias-unet.yaml", "debias-text-encoder.yaml", "debias-text-encoder-and-unet.yaml"]:
        path = tmp_path / f
        path.write_text(yaml.safe_dump(GOLD_CLI[f]["yaml"]))
        a = vars(parse_args(["--config", str(path)]))
        a["config"] = f
        assert a == GOLD_CLI[f]["args"], f
    os.environ["LOCAL_RANK"] = "3"
    try:
        assert parse_args([]).local_rank == 3
    finally:
        os.environ.pop("LOCAL_RANK")


# ------------------------------------------------------------------ parameter inventories == oracle modules (drop-in key contract)
def test_param_shapes_match_oracle_modules():
    from finetune_fair_diffusion_amd import weights as W
    from oracle import nn_clip, nn_mobilenet, nn_unet, nn_vae
    with torch.device("meta"):
        u = nn_unet.UNet2DConditionModel(nn_unet.UNetConfig())
        v = nn_vae.AutoencoderKLDecoder()
        c = nn_clip.CLIPTextModel()
        m = nn_mobilenet.MobileNetV3Large(80)
        lora = nn_unet.make_unet_lora(u, 50)
    for mod, spec in [(u, {k: s for k, s in W.unet_param_shapes(W.UNetConfig()).items()}), (v, W.vae_param_shapes(W.VAEConfig())),
                      (c, W.clip_param_shapes(W.CLIPTextConfig())), (m, W.mobilenet_param_shapes(80))]:
        sd = {k: tuple(t.shape) for k, t in mod.state_dict().items() if ".processor." not in k}
        assert sd == {k: tuple(s) for k, s in spec.items()}
    ls = {k: tuple(t.shape) for k, t in lora.state_dict().items()}
    sp = {k: tuple(s) for k, s in W.unet_lora_param_shapes(W.UNetConfig(), 50).items()}
    assert ls == sp and len(ls) == 256
    assert sum(int(np.prod(s)) for s in sp.values()) == 9_964_800  # r=50: 9.96 M params
    assert sum(p.numel() for p in u.parameters() if p.requires_grad) - sum(int(np.prod(s)) for s in sp.values()) == 859_520_964
    te_sp = W.clip_lora_param_shapes(W.CLIPTextConfig(), 50)
    assert len(te_sp) == 144 and sum(int(np.prod(s)) for s in te_sp.values()) == 8_294_400  # 8.29 M


# ------------------------------------------------------------------ scheduler analytics (oracle and product agree; exactness on a toy ODE)
def test_scheduler_oracle_vs_product_coefficients_and_chain():
    from finetune_fair_diffusion_amd.scheduler import DPMSolverMultistepScheduler as P
    from oracle.dpm_solver import DPMSolverMultistepScheduler as O
    from oracle.fair_step import grad_coefs
    for S in (4, 19, 20, 23, 50):
        o, p = O(), P()
        o.set_timesteps(S)
        p.set_timesteps(S)
        assert torch.equal(o.timesteps, p.timesteps)
        if S == 20:
            assert o.timesteps.tolist()[:3] == [999, 949, 899] and o.timesteps.tolist()[-1] == 50
        # scalar chain: autograd through the oracle scheduler == product's closed recurrence
        eps = [torch.zeros(1, requires_grad=True) for _ in range(S)]
        lat = torch.ones(1)
        for i, t in enumerate(o.timesteps):
            lat = o.step(eps[i], t, lat).prev_sample
        lat.backward()
        assert np.allclose([e.grad.item() for e in eps], p.chain_coefs(), rtol=1e-4, atol=1e-6)
        assert np.allclose(grad_coefs(o), p.grad_coefs(), rtol=1e-6)
        assert abs(np.prod(p.grad_coefs()) - 1.0) < 1e-6  # normalised by the geometric mean (:1109)


def test_dpm_solver_exact_on_gaussian_toy():
    """For data ~ N(0, s^2 I) the optimal eps is linear in x and x0-prediction DPM-Solver++ converges to the
    exact marginal-preserving map as S grows: x_0 = x_T * s / sqrt(alpha_T^2 s^2 + sigma_T^2)."""
    from oracle.dpm_solver import DPMSolverMultistepScheduler as O
    s = 0.7
    errs = []
    for S in (10, 40, 160):
        o = O()
        o.set_timesteps(S)
        x = torch.tensor([1.3])
        for t in o.timesteps:
            a, sg = o.alpha_t[t], o.sigma_t[t]
            eps = sg * x / (a * a * s * s + sg * sg)   # E[eps | x_t] for Gaussian data
            x = o.step(eps, t, x).prev_sample
        aT, sT = o.alpha_t[999], o.sigma_t[999]
        a0, s0 = o.alpha_t[0], o.sigma_t[0]
        exact = 1.3 * math.sqrt((a0 * a0 * s * s + s0 * s0) / (aT * aT * s * s + sT * sT))
        errs.append(abs(float(x) - exact))
    assert errs[2] < errs[1] < errs[0] and errs[2] < 5e-3 and errs[2] < errs[0] / 10


def test_ema_schedule_matches_oracle():
    from finetune_fair_diffusion_amd.step import EMAState
    from oracle.fair_step import EMAModel
    p = [torch.zeros(3)]
    o, e = EMAModel(p, decay=0.996), EMAState(0.996)
    for n in range(1, 400):
        o.optimization_step = n - 1
        o.optimization_step += 1
        assert abs((1 - o.get_decay(n)) - e.next_one_minus_decay()) < 1e-12


def test_oracle_invariants_zero_lora_and_cfg():
    """Zero-initialised LoRA ``up`` leaves the U-Net output unchanged; guidance 1 returns the conditional branch."""
    import util_models as U
    from oracle import nn_unet
    om = U.oracle_models(train_unet=False, train_te=False)
    unet = om["unet"]
    x, enc = torch.randn(2, 4, 32, 32), torch.randn(2, 7, 64)
    with torch.no_grad():
        y0 = unet(x, torch.tensor(500), enc).sample
        nn_unet.make_unet_lora(unet, 4)  # up = 0
        y1 = unet(x, torch.tensor(500), enc).sample
    assert torch.equal(y0, y1)


# ------------------------------------------------------------------ C-ABI library: loads and exports every declared symbol
def test_gemm_dispatch_name_statistics_rows_and_launcher_agree():
    """fd_gemm, fd_gemm_kernel_name and fd_gemm_stats_rows share one plan (launcher and name function had drifted apart).  Host-only
    functions: the kernel each of the step's shape families gets, its split-K factor, and which of them can write GroupNorm statistics."""
    from finetune_fair_diffusion_amd import lib
    L = lib.load()

    def plan(M, N, K, conv=None, stats=False, **kw):
        d = lib.GemmDesc()
        d.M, d.N, d.K, d.batch, d.ldc, d.lda, d.ldb, d.alpha = M, N, K, 1, N, K, K, 1.0
        d.workspace, d.workspace_bytes = 1 << 20, 64 << 20          # a non-null workspace enables split-K in the policy (never dereferenced here)
        if conv is not None:
            B, H, Cin, mode = conv
            d.conv, d.conv_mode, d.Bn, d.H, d.W, d.Cin = 1, mode, B, H, H, Cin
            d.Ho = d.Wo = H // 2 if mode == 1 else (2 * H if mode == 2 else H)        # the phase pairs (4, 6) count low-res rows per phase
        for k, v in kw.items():
            setattr(d, k, v)
        if stats:
            d.gn_stats = 1 << 20
        buf = ctypes.create_string_buffer(128)
        split = L.fd_gemm_kernel_name(ctypes.byref(d), buf, 128)
        return buf.value.decode(), split, L.fd_gemm_stats_rows(ctypes.byref(d))

    assert plan(65536, 320, 320) == ("gemm_big_kernel<256, 320, 4, 4, 0>", 0, 32)
    assert plan(65536, 320, 320, stats=True) == ("gemm_big_kernel<256, 320, 4, 4, 3>", 0, 32)
    assert plan(4096, 1280, 1280, stats=True) == ("gemm_big_kernel<128, 320, 4, 4, 3>", 0, 32)
    assert plan(65536, 2560, 320, act=7) == ("gemm_pp_kernel<256, 0, true>", 0, 0)                           # FF1 / GEGLU: ping-pong, no statistics
    # stride-1 3x3 convolutions of the 64^2 / 32^2 / 16^2 levels: the halo-staged kernel (round 6); other geometries keep the per-tap ping-pong kernel
    assert plan(65536, 320, 2880, conv=(16, 64, 320, 0), stats=True) == ("conv_halo_kernel<256, 64, 3, true>", 0, 32)
    assert plan(16384, 640, 5760, conv=(16, 32, 640, 0)) == ("conv_halo_kernel<256, 32, 1, true>", 0, 32)
    assert plan(4096, 1280, 11520, conv=(16, 16, 1280, 0)) == ("conv_halo_kernel<128, 16, 1, true>", 0, 32)
    assert plan(2 * 24 * 24 * 40, 320, 2880, conv=(80, 24, 320, 0), stats=True) == ("gemm_pp_kernel<256, 3, true>", 0, 32)      # 24 x 24 maps: no halo geometry
    assert plan(16384, 320, 2880, conv=(16, 64, 320, 1), stats=True) == ("gemm_big_kernel<128, 320, 4, 4, 4>", 0, 32)     # stride 2: lockstep gather
    name, split, rows = plan(1024, 1280, 11520, conv=(16, 8, 1280, 0))                                        # 8^2 level: split-K, epilogue in the reduce kernel
    assert name == "gemm_big_kernel<128, 320, 4, 4, 1>" and split == 8 and rows == 0
    assert plan(65536, 8, 320)[0].startswith("gemm_skinny_kernel<1, 1") and plan(65536, 8, 320)[2] == 0
    assert plan(51300, 512, 1096) == ("gemm_big_kernel<256, 256, 2, 4, 0>", 0, 0)                             # 64-column wave tiles: no statistics epilogue
    assert plan(300, 320, 320) == ("gemm_glds_kernel<64, 64, false>", 0, 0)
    # the up-sampling phase pair writing the channels-last result (FD_CONV_UP2PI = 6): statistics epilogue as instantiation <..., 6>; the phase-major form has none
    assert plan(16384, 640, 4 * 640, conv=(16, 32, 640, 6), stats=True) == ("gemm_big_kernel<256, 320, 2, 4, 6>", 0, 32)
    assert plan(16384, 640, 4 * 640, conv=(16, 32, 640, 4)) == ("gemm_big_kernel<256, 320, 2, 4, 2>", 0, 0)
    # the one-launch cross-attention sub-block validates its shapes on the host
    c = lib.CrossBlockDesc()
    for f in ("x", "ln2_gamma", "ln2_beta", "wq", "k", "vt", "wo", "bo", "y"):
        setattr(c, f, 1 << 20)
    c.M, c.C, c.heads, c.rows_per_sample, c.kv_div, c.L, c.Lp, c.scale = 4096, 320, 8, 1024, 1, 77, 80, 40 ** -0.5
    for field, bad, msg in (("C", 1280, b"C=1280"), ("L", 81, b"<= 80 keys"), ("M", 4100, b"multiples of the 64-row tile"), ("heads", 5, b"heads=5"), ("Lp", 72, b"Lp=72")):
        good = getattr(c, field)
        setattr(c, field, bad)
        assert L.fd_cross_attn_block(ctypes.byref(c), None) == -1 and msg in L.fd_last_error(), (field, L.fd_last_error())
        setattr(c, field, good)
    c.struct_size = 8
    assert L.fd_cross_attn_block(ctypes.byref(c), None) == -1 and b"struct_size" in L.fd_last_error()
    # fd_gemm refuses gn_stats where the plan has no statistics epilogue (validated on the host, before any launch)
    d = lib.GemmDesc()
    d.A = d.B = d.C = d.gn_stats = 1 << 20
    d.M, d.N, d.K, d.batch, d.ldc, d.lda, d.ldb = 300, 320, 320, 1, 320, 320, 320
    assert L.fd_gemm(ctypes.byref(d), None) == -1 and b"no statistics epilogue" in L.fd_last_error()


def _header_struct_fields(name):
    """[(field, ctypes type)] of ``typedef struct <name> { ... }`` in include/fairdiff_hip.h, in declaration order."""
    from finetune_fair_diffusion_amd import lib
    src = re.sub(r"/\*.*?\*/", " ", open(lib.HEADER_PATH).read(), flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), src, flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        if "*" in decl:                                   # ``const void* A`` / ``float* gn_stats`` / ``const float* down``
            names, ct = [decl.split("*")[-1]], ctypes.c_void_p
        else:                                             # ``int32_t M, N, K, K2`` / ``int64_t lda`` / ``float alpha``
            t, rest = decl.split(" ", 1)
            names, ct = rest.split(","), {"int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "float": ctypes.c_float}[t]
        out += [(n.strip(), ct) for n in names]
    return out


def test_integration_md_descriptor_mirror_matches_header_and_lib(tmp_path):
    """VERDICT r4 row b: the reference-side stub documented in INTEGRATION.md had gone stale against the header (8 bytes short).  Three statements of
    fd_gemm_desc must agree field for field -- the header (as gcc lays it out), lib.GemmDesc, and the ctypes class a maintainer copies out of
    INTEGRATION.md -- likewise the two other descriptor structs, and every entry point refuses a descriptor whose struct_size is not its own sizeof."""
    import subprocess
    from finetune_fair_diffusion_amd import lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    snippet = re.search(r"(class fd_gemm_desc\(ctypes\.Structure\):.*?\])\s*#[^\n]*\nassert L\.fd_version\(\) == (\d+)", md, flags=re.S)
    assert snippet, "INTEGRATION.md: the fd_gemm_desc mirror (followed by the fd_version assert) is gone"
    ns = {"ctypes": ctypes}
    exec(snippet.group(1), ns)
    doc_fields = ns["fd_gemm_desc"]._fields_
    assert int(snippet.group(2)) == lib.ABI_VERSION
    assert doc_fields == lib.GemmDesc._fields_ == _header_struct_fields("fd_gemm_desc")
    assert doc_fields[0] == ("struct_size", ctypes.c_int32)
    assert "struct_size=ctypes.sizeof(fd_gemm_desc)" in md, "INTEGRATION.md's example must fill struct_size"
    assert lib.WgradDesc._fields_ == _header_struct_fields("fd_wgrad_desc") and lib.LoraRefreshDesc._fields_ == _header_struct_fields("fd_lora_refresh_desc")
    assert lib.CrossBlockDesc._fields_ == _header_struct_fields("fd_cross_block_desc")
    # the compiler's own layout: sizeof and every offset
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "fairdiff_hip.h"', 'int main(void) {']
    for cname, cls in (("fd_gemm_desc", lib.GemmDesc), ("fd_wgrad_desc", lib.WgradDesc), ("fd_lora_refresh_desc", lib.LoraRefreshDesc),
                       ("fd_cross_block_desc", lib.CrossBlockDesc)):
        prog.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f, _ in cls._fields_:
            prog.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    prog += ['printf("version %d\\n", FD_ABI_VERSION);', 'return 0; }']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(prog))
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(c), "-o", str(tmp_path / "layout")], check=True)
    got = dict(line.rsplit(" ", 1) for line in subprocess.run([str(tmp_path / "layout")], check=True, capture_output=True, text=True).stdout.splitlines())
    assert int(got["version"]) == lib.ABI_VERSION
    for cname, cls in (("fd_gemm_desc", lib.GemmDesc), ("fd_wgrad_desc", lib.WgradDesc), ("fd_lora_refresh_desc", lib.LoraRefreshDesc),
                       ("fd_cross_block_desc", lib.CrossBlockDesc)):
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, f)]) == getattr(cls, f).offset, (cname, f)
    # the library checks it (host-side, before any launch)
    L = lib.load()
    assert L.fd_version() == lib.ABI_VERSION
    d = lib.GemmDesc()
    assert d.struct_size == ctypes.sizeof(lib.GemmDesc)
    d.A = d.B = d.C = 1 << 20
    d.M, d.N, d.K, d.batch, d.ldc, d.lda, d.ldb = 300, 320, 320, 1, 320, 320, 320
    buf = ctypes.create_string_buffer(128)
    for bad in (0, ctypes.sizeof(lib.GemmDesc) - 8, 304):          # 304: the round-4 layout a stale binding would claim
        d.struct_size = bad
        for fn, args in (("fd_gemm", (None,)), ("fd_gemm_tile", ()), ("fd_gemm_stats_rows", ()), ("fd_gemm_kernel_name", (buf, 128))):
            assert getattr(L, fn)(ctypes.byref(d), *args) == -1 and b"struct_size" in L.fd_last_error(), (fn, bad)
    w = lib.WgradDesc.array(2)
    assert w[1].struct_size == ctypes.sizeof(lib.WgradDesc)
    w[1].struct_size = 0
    assert L.fd_lora_wgrad_multi(ctypes.byref(w), 2, None, 0, None) == -1 and b"struct_size" in L.fd_last_error()
    r = (lib.LoraRefreshDesc * 1)()                                 # constructed without the helper: struct_size stays 0
    assert L.fd_lora_refresh_multi(ctypes.byref(r), 1, None) == -1 and b"struct_size" in L.fd_last_error()


def test_library_exports_every_header_symbol():
    from finetune_fair_diffusion_amd import lib
    protos = lib.parse_header()
    assert len(protos) >= 40 and "fd_gemm" in protos and "fd_attn_fwd" in protos
    L = lib.load()
    for name in protos:
        assert hasattr(L, name), name
    assert L.fd_version() == lib.ABI_VERSION == 4
    assert ctypes.sizeof(lib.GemmDesc) == 264  # keep the Python mirror in step with fd_gemm_desc (gcc's own sizeof / offsets: the test below)
    # argument validation happens on the host before any launch: safe to exercise without a GPU
    d = lib.GemmDesc()
    assert L.fd_gemm(ctypes.byref(d), None) == -1 and b"null operand" in L.fd_last_error()
    assert L.fd_layernorm_fwd(None, None, None, None, None, 4, 12, 1e-5, None) == -1
    # "pre-scaled q" (negative softmax scale) is refused on the host where the head dim has no spare contraction slots or a transposed-copy form is asked for
    one = ctypes.c_void_p(1 << 20)
    assert L.fd_attn_fwd(one, one, one, one, None, 2, 8, 256, 256, 256, 80, 1, -0.1118, 0, 0, None) == -1 and b"pre-scaled q" in L.fd_last_error()
    assert L.fd_attn_bwd_dq(one, one, one, one, one, one, one, one, 2, 8, 256, 256, 256, 80, 1, -0.1118, 0, 0, 0, None) == -1 and b"pre-scaled q" in L.fd_last_error()
    assert L.fd_attn_bwd_dkdv(one, one, one, one, one, one, one, one, 2, 8, 256, 256, 256, 80, 1, -0.1118, 0, 0, 0, 0, None) == -1
    d = lib.GemmDesc()
    d.A = d.B = d.C = 1 << 20
    d.M, d.N, d.K, d.batch, d.ldc, d.lda, d.ldb, d.colscale_cols = 300, 320, 320, 1, 320, 320, 320, 6
    assert L.fd_gemm(ctypes.byref(d), None) == -1 and b"colscale_cols" in L.fd_last_error()
    assert L.fd_working_dtype() == b"fp16"
    assert not hasattr(L, "fd_attn_fwd_fp8")          # the e4m3 self-attention left the product in round 5 (scratch/attn_fp8_experiment.hip)
    # the bf16 build (BASELINE configs[4]) of the same sources exports the same C-ABI
    path = os.path.join(os.path.dirname(lib.LIB_PATH), "libfairdiff_hip_bf16.so")
    assert os.path.exists(path), "libfairdiff_hip_bf16.so missing: run __graft_entry__.build()"
    Lb = ctypes.CDLL(path)
    for name in protos:
        assert hasattr(Lb, name), name
    Lb.fd_working_dtype.restype = ctypes.c_char_p
    assert Lb.fd_working_dtype() == b"bf16"
    # a library of the wrong working dtype is refused, not silently used
    import subprocess, sys
    r = subprocess.run([sys.executable, "-c", "from finetune_fair_diffusion_amd import lib; lib.load()"],
                       env=dict(os.environ, FD_DTYPE="bf16", FAIRDIFF_LIB=lib.LIB_PATH), capture_output=True, text=True, cwd=os.path.dirname(HERE))
    assert r.returncode != 0 and "was built for fp16" in r.stderr


def test_working_dtype_preselection_from_the_reference_flag(tmp_path):
    """``--mixed_precision bf16`` (exp-1 main:401-405) or a --config YAML carrying it selects the bf16 library before the package binds its
    dtype, with ``parse_args``' precedence (the YAML overlay is applied last, :625-642); an explicit FD_DTYPE wins; a host program that
    merely imports the package never has its own argv parsed."""
    import subprocess, sys
    code = ("import sys, finetune_fair_diffusion_amd as P; P._preselect_working_dtype(sys.argv); "
            "from finetune_fair_diffusion_amd import lib; print(lib.WORKING_DTYPE, lib.LIB_PATH.rsplit('/', 1)[1])")
    env = {k: v for k, v in os.environ.items() if k not in ("FD_DTYPE", "FAIRDIFF_LIB")}
    run = lambda argv, e=env, c=code: subprocess.run([sys.executable, "-c", c] + argv, env=e, capture_output=True, text=True, cwd=os.path.dirname(HERE)).stdout.split()  # noqa: E731
    assert run([]) == ["fp16", "libfairdiff_hip.so"]
    assert run(["--mixed_precision", "bf16"]) == ["bf16", "libfairdiff_hip_bf16.so"]
    assert run(["--mixed_precision=bf16"])[0] == "bf16"
    y = tmp_path / "c.yaml"
    y.write_text("mixed_precision: bf16\nrank: 4\n")
    assert run(["--config", str(y)])[0] == "bf16"
    assert run(["--config", str(y), "--mixed_precision", "fp16"])[0] == "bf16"          # the YAML wins, as in parse_args
    from finetune_fair_diffusion_amd.cli import parse_args
    assert parse_args(["--config", str(y), "--mixed_precision", "fp16"]).mixed_precision == "bf16"
    assert run(["--mixed_precision", "bf16"], dict(env, FD_DTYPE="fp16"))[0] == "fp16"
    # a plain import from a host program leaves that program's argv alone
    plain = "from finetune_fair_diffusion_amd import lib; print(lib.WORKING_DTYPE)"
    assert run(["--mixed_precision", "bf16"], env, plain) == ["fp16"]
    # the package's own entry point does read it (python -m ...train): flag fp16 + YAML bf16 -> bf16 on both sides, so the driver gets past
    # its dtype check and stops at the device check on a CPU-only box
    if not torch.cuda.is_available():
        for mflag in ("-m", "-um"):          # combined short flags too
            r = subprocess.run([sys.executable, mflag, "finetune_fair_diffusion_amd.train", "--synthetic", "--config", str(y), "--mixed_precision", "fp16"],
                               env=env, capture_output=True, text=True, cwd=os.path.dirname(HERE))
            assert r.returncode != 0 and "needs an MI355X" in r.stderr and "was started with" not in r.stderr, r.stderr[-600:]


def test_no_packed_fp32_valu_in_shipped_code_objects():
    """Build invariant behind the round-3 hazard's fix: packed-fp32 VALU sequences (v_pk_add / mul / fma_f32, produced by
    hipcc's SLP pass and by <2 x float> instruction selection) returned wrong lanes on gfx950 when another stream's kernel shared the SIMD, so
    none may appear in any kernel of either shipped library.  Also pins the scratch use of the kernels the dispatch can reach."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "finetune_fair_diffusion_amd", "csrc"))
    import codeobj
    pkg = os.path.join(os.path.dirname(HERE), "finetune_fair_diffusion_amd")
    for lib in ("libfairdiff_hip.so", "libfairdiff_hip_bf16.so"):
        path = os.path.join(pkg, lib)
        sites = codeobj.packed_f32_sites(path)
        assert not sites, f"{lib}: {len(sites)} packed-fp32 VALU instructions, e.g. {sites[:3]}"
        nk = 0
        for _, co in codeobj.code_objects(path):
            for name, r in codeobj.kernel_resources(co).items():
                nk += 1
                if "layernorm_kernel" in name or "attn_fwd_kernel" in name or "gemm_pp_kernel" in name or "attn_bwd_dkdv" in name:
                    assert r["scratch"] == 0, (lib, name, r)
        assert nk > 150, (lib, nk)


def test_device_side_exp1_loss_assembly_equals_the_host_statement():
    """fairness_dev.py (what the step runs ON THE DEVICE for exp-1 since round 4: ranks -> binomial targets / uncertainties -> threshold, cross-entropy
    with -1 sentinels and its logit gradient, dynamic weights, hook factors) against fairness.py, the host restatement that the
    reference-executed goldens pin -- on CPU tensors, 300 random batches of 1..24 images with missing faces."""
    from finetune_fair_diffusion_amd import fairness as FH, fairness_dev as FD
    g = torch.Generator().manual_seed(0)
    tabs = FD.binomial_tables(24)
    for trial in range(300):
        n = int(torch.randint(1, 25, (1,), generator=g))
        p1 = torch.rand(n, generator=g)
        probs = torch.stack([1 - p1, p1], -1)
        nf = torch.rand(n, generator=g) < 0.2
        probs[nf] = -1
        t_ref, u_ref = FH.generate_dynamic_targets(probs, w_uncertainty=True)
        t, u = FD.dynamic_targets(probs, tabs)
        assert torch.equal(t, t_ref) and torch.equal(u, u_ref), (trial, t, t_ref)
        thr = float(torch.rand(1, generator=g)) * 0.5
        t2 = t_ref.clone()
        t2[u_ref > thr] = -1
        assert torch.equal(FD.dynamic_targets(probs, tabs, threshold=thr)[0], t2)
        face, logits, w = ~nf, torch.randn(n, 2, generator=g) * 3, torch.rand(n, generator=g)
        l_ref, d_ref = FH.fair_loss_and_grad(logits, t2, face, w)
        l, d = FD.fair_loss_and_grad(logits, t2, face, w)
        assert torch.allclose(l, l_ref, atol=1e-6) and torch.allclose(d, d_ref, atol=1e-6) and torch.equal(l == -1, l_ref == -1)
        po = torch.randint(-1, 2, (n,), generator=g)
        assert torch.equal(FD.dynamic_weights(face, t2, po, 0.2), FH.gen_dynamic_weights(face, t2, po, 0.2))
        boxes = torch.randint(0, 60, (n, 4), generator=g).int()
        boxes[nf] = -1
        _, f_ref = FH.face_grad_factors(boxes, boxes, t2, po, 0.3, 64, 64)
        assert torch.equal(FD.hook_factors(~(boxes == -1).all(dim=1), t2, po, 0.3), f_ref)
        sel = face.nonzero().view(-1)
        pr, pd, lg = FD.probs_preds(logits[sel], sel, n, 0, 2)
        assert torch.equal(pr == -1, probs == -1) and torch.equal(pd[sel], torch.softmax(logits[sel], -1).max(-1).indices) and bool((pd[nf] == -1).all())


def test_product_fails_loudly_without_library(monkeypatch):
    from finetune_fair_diffusion_amd import lib
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libfairdiff_hip.so")
    with pytest.raises(RuntimeError, match="no fallback"):
        lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(os.path.dirname(HERE), "finetune_fair_diffusion_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "import oracle" not in src and "from oracle" not in src, f


# ------------------------------------------------------------------ multi-attribute OT targets (exp-3 / exp-4): product (assignment) vs oracle (LP)
@pytest.mark.parametrize("exp", ["exp-3", "exp-4"])
def test_multi_attribute_targets_product_vs_oracle(exp):
    from finetune_fair_diffusion_amd.fairness import EXPERIMENT_ATTRS, generate_dynamic_targets_multi as prod
    from oracle.fair_step import generate_dynamic_targets_multi as orc
    _, attrs, cdfs, asym = EXPERIMENT_ATTRS[exp]
    g = torch.Generator().manual_seed(3)
    n = 12
    probs = [torch.softmax(torch.randn(n, k, generator=g) * 2, -1) for _, _, k in attrs]
    for p in probs:
        p[5] = -1  # a missing face
    rp, tp_p = prod(probs, cdfs, 25, torch.Generator().manual_seed(9), None, asym, return_plan=True)
    (ro, tp) = orc(probs, cdfs, 25, torch.Generator().manual_seed(9), asym)
    assert abs(float(tp.sum(dim=1).max()) - 1) < 1e-5
    # identical Monte-Carlo draws, two independent solvers (assignment on the sink-replicated costs vs the transport LP): with
    # generic (random, tie-free) costs the optimal plan of every draw is unique, so the AVERAGED PLANS agree to rounding -- and with
    # them every marginal, target and uncertainty (compare tp itself, not loose uncertainties)
    assert tp_p.shape == tp.shape and float((tp_p - tp).abs().max()) < 1e-6, float((tp_p - tp).abs().max())
    for (tpd, upd), (tor, uor) in zip(rp, ro):
        assert tpd[5] == -1 and tor[5] == -1
        assert np.abs(upd.numpy() - uor.numpy()).max() < 1e-5
        assert torch.equal(tpd, tor)


# ------------------------------------------------------------------------------------------ CLI of exp-3/4/5, lr schedule, checkpoints
def test_cli_multi_attribute_experiments_match_reference(tmp_path):
    """parse_args of exp-2/3/4/5 (defaults + every YAML overlay) against the reference's own parse_args output."""
    import yaml
    from finetune_fair_diffusion_amd.cli import parse_args
    gold = json.load(open(os.path.join(HERE, "golden", "reference_cli_multi.json")))
    assert set(gold) == {"exp-2", "exp-3", "exp-4", "exp-5"}
    for exp, cases in gold.items():
        assert vars(parse_args([], experiment=exp)) == cases["defaults"]
        for f, c in cases.items():
            if f == "defaults":
                continue
            p = tmp_path / f"{exp}-{f}"
            p.write_text(yaml.safe_dump(c["yaml"]))
            d = vars(parse_args(["--config", str(p)], experiment=exp))
            d["config"] = f
            assert d == c["args"], (exp, f)


def test_lr_schedule_matches_lambda_lr():
    """lr multipliers vs the schedule functions the reference's get_scheduler wraps (same formulas ship in transformers)."""
    import transformers.optimization as topt
    from finetune_fair_diffusion_amd.lr_schedule import lr_lambda
    w, T, base = 3, 20, 5e-5

    def run(make):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=base)
        sch = make(opt)
        out = []
        for _ in range(T + 4):
            out.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        return out
    cases = {
        "constant": lambda o: topt.get_constant_schedule(o),
        "constant_with_warmup": lambda o: topt.get_constant_schedule_with_warmup(o, w),
        "linear": lambda o: topt.get_linear_schedule_with_warmup(o, w, T),
        "cosine": lambda o: topt.get_cosine_schedule_with_warmup(o, w, T),
        "cosine_with_restarts": lambda o: topt.get_cosine_with_hard_restarts_schedule_with_warmup(o, w, T, num_cycles=3),
        "polynomial": lambda o: topt.get_polynomial_decay_schedule_with_warmup(o, w, T, power=2.0),
    }
    for name, make in cases.items():
        ref = run(make)
        mine = [base * lr_lambda(name, s, w, T, 3, 2.0, base) for s in range(T + 4)]
        assert np.allclose(ref, mine, rtol=1e-9, atol=1e-15), name
    with pytest.raises(ValueError):
        lr_lambda("nope", 0)


class _FakeModel:
    def __init__(self, bank):
        self.lora_bank = bank
        self.refreshed = 0

    def refresh_lora(self):
        self.refreshed += 1


def _fake_trainer(seed):
    import types
    from finetune_fair_diffusion_amd.layers import ParamBank
    from finetune_fair_diffusion_amd.step import EMAState
    from finetune_fair_diffusion_amd import weights as W
    from util_models import TINY_UNET, TINY_CLIP
    TINY_UNET, TINY_CLIP = W.UNetConfig(**TINY_UNET), W.CLIPTextConfig(**TINY_CLIP)
    g = torch.Generator().manual_seed(seed)
    ub = ParamBank(W.unet_lora_param_shapes(TINY_UNET, 4), "cpu")
    tb = ParamBank(W.clip_lora_param_shapes(TINY_CLIP, 4), "cpu")
    for b in (ub, tb):
        for buf in (b.flat, b.exp_avg, b.exp_avg_sq, b.ema):
            buf.copy_(torch.randn(buf.shape, generator=g))
    t = types.SimpleNamespace(args=types.SimpleNamespace(train_unet=True, train_text_encoder=True), unet=_FakeModel(ub), te=_FakeModel(tb),
                              ema=[EMAState(0.996), EMAState(0.996)], opt_step=seed, lr_step=seed + 1,
                              target_rng=torch.Generator().manual_seed(seed))
    t.ema[0].optimization_step = 7 + seed
    return t


def test_detector_face_provider_matches_reference_get_face_on_scripted_detections():
    """The adaptor behind the provider seam vs the reference's own get_face / get_face_app / get_face_FR (:1192-1353) replaying the same
    scripted detector outputs (tests/golden/reference_face_provider.json, made by make_golden.face_provider_golden): which face wins, box
    order and expansion, the five landmarks, fill values, and that face_recognition is consulted only where insightface found nothing."""
    import types
    from finetune_fair_diffusion_amd.fairness import DetectorFaceProvider
    gold = json.load(open(os.path.join(HERE, "golden", "reference_face_provider.json")))
    assert len(gold) == 6
    n_fr_calls = 0
    for case in gold:
        N = len(case["app"])
        images = torch.zeros(N, 3, case["H"], case["W"])
        app_q = list(case["app"])
        fr_q = [f for f, a in zip(case["fr"], case["app"]) if len(a) == 0]
        state = {}

        def app_get(img_bgr):
            assert img_bgr.dtype == np.uint8 and img_bgr.shape == (case["H"], case["W"], 3)
            return [dict(bbox=np.array(d["bbox"]), kps=np.array(d["kps"])) for d in app_q.pop(0)]

        def fr_locations(img, model, number_of_times_to_upsample):
            assert model == "cnn" and number_of_times_to_upsample == 0
            state["cur"] = fr_q.pop(0)
            return [tuple(l) for l in state["cur"]["locations"]]

        def fr_landmarks(img, face_locations, model):
            assert model == "large" and len(face_locations) == 1
            return [state["cur"]["landmarks"][[tuple(l) for l in state["cur"]["locations"]].index(tuple(face_locations[0]))]]
        prov = DetectorFaceProvider(types.SimpleNamespace(get=app_get), types.SimpleNamespace(face_locations=fr_locations, face_landmarks=fr_landmarks))
        ind, boxes = prov(images)
        lms = prov.landmarks(images)              # same images object: detectors are not run a second time
        assert not app_q and not fr_q
        n_fr_calls += sum(len(a) == 0 for a in case["app"])
        assert ind.tolist() == case["indicators"] and boxes.dtype == torch.int32 and boxes.tolist() == case["boxes"]
        assert torch.allclose(lms, torch.tensor(case["landmarks"]), atol=1e-4)
    assert n_fr_calls > 0 and any(not all(c["indicators"]) for c in gold)
    with pytest.raises(ValueError):
        DetectorFaceProvider()
    with pytest.raises(ImportError):
        DetectorFaceProvider.from_installed()     # the detector packages are not part of this image


def test_prefix_embedding_checkpoint_files_are_fair_embeddings_state_dicts(tmp_path):
    """exp-2: the trained table [n+1, D] (row 0 zero, rows 1..n initialised from existing vocabulary rows, exp-2 1-main-debias.py:86-146)
    and its checkpoint files in the FairEmbeddings state-dict format of exp-2's 2-export-checkpoint.py:566-575."""
    import types
    from finetune_fair_diffusion_amd import checkpoint as ck, generate
    from finetune_fair_diffusion_amd.prefix import PrefixEmbedding
    from finetune_fair_diffusion_amd.step import EMAState
    tok = torch.randn(50, 16, generator=torch.Generator().manual_seed(1)).half()
    te = types.SimpleNamespace(tok=tok, pos=torch.randn(9, 16).half())
    pe = PrefixEmbedding(te, 4, "cpu", seed=5)
    assert pe.weight.shape == (5, 16) and float(pe.weight[0].abs().max()) == 0
    for r in pe.weight[1:]:                                   # every prefix vector is a copy of one vocabulary row
        assert any(torch.equal(r, v.float()) for v in tok)
    assert torch.equal(PrefixEmbedding(te, 4, "cpu", seed=5).weight, pe.weight) and not torch.equal(PrefixEmbedding(te, 4, "cpu", seed=6).weight, pe.weight)
    pe.bank.ema.mul_(0.5)
    pe.bank.exp_avg.fill_(0.25)
    tr = types.SimpleNamespace(args=types.SimpleNamespace(), prefix=pe, ema=[EMAState(0.996)], opt_step=3, lr_step=3,
                               target_rng=torch.Generator().manual_seed(1), unet=None, te=None)
    path = ck.save_state(tr, str(tmp_path / "checkpoint-3"), 3)
    assert sorted(os.listdir(path)) == ["prefix_embedding.pth", "prefix_embedding_EMA.pth", "rng_rank0.pth", "trainer_state.pth"]
    sd, sde = torch.load(os.path.join(path, "prefix_embedding.pth")), torch.load(os.path.join(path, "prefix_embedding_EMA.pth"))
    assert sorted(sd) == ["position_embedding.weight", "position_ids", "token_embedding.weight"] and sd["position_ids"].shape == (1, 9)
    assert torch.equal(sd["token_embedding.weight"], pe.weight) and torch.equal(sde["token_embedding.weight"], 0.5 * pe.weight)
    assert torch.equal(generate.load_prefix_embedding(os.path.join(path, "prefix_embedding.pth"), 4), pe.vectors())
    with pytest.raises(ValueError):
        generate.load_prefix_embedding(os.path.join(path, "prefix_embedding.pth"), 5)
    other = PrefixEmbedding(te, 4, "cpu", seed=9)
    tr2 = types.SimpleNamespace(args=types.SimpleNamespace(), prefix=other, ema=[EMAState(0.996)], opt_step=0, lr_step=0,
                                target_rng=torch.Generator().manual_seed(2), unet=None, te=None)
    assert ck.load_state(tr2, path) == 3 and tr2.opt_step == 3
    for buf in ("flat", "ema", "exp_avg", "exp_avg_sq"):
        assert torch.equal(getattr(other.bank, buf), getattr(pe.bank, buf)), buf
    # a LoRA checkpoint cannot be resumed as a prefix run
    with pytest.raises((ValueError, FileNotFoundError, KeyError)):
        ck.load_state(tr2, str(tmp_path / "nope"))


def test_checkpoint_round_trip_export_and_rolling_cleanup(tmp_path):
    """Trainer state -> checkpoint_tmp-N -> fresh trainer; the four exported files are the reference's public format
    (2-export-checkpoint.py:619-642): fp32 CPU dict[str,Tensor] keyed by the diffusers LoRA names."""
    from finetune_fair_diffusion_amd import checkpoint as ck
    a, b = _fake_trainer(1), _fake_trainer(2)
    d = str(tmp_path / "checkpoints")
    os.makedirs(d)
    a.target_rng.manual_seed(99)
    torch.manual_seed(123)
    path = ck.save_state(a, os.path.join(d, "checkpoint_tmp-20"), 20)
    expect_next = torch.rand(3)
    assert sorted(os.listdir(path)) == ["rng_rank0.pth", "text_encoder_lora.pth", "text_encoder_lora_EMA.pth", "trainer_state.pth", "unet_lora.pth", "unet_lora_EMA.pth"]
    sd = torch.load(os.path.join(path, "unet_lora.pth"))
    k0 = "down_blocks.0.attentions.0.transformer_blocks.0.attn1.processor.to_q_lora.down.weight"
    assert k0 in sd and sd[k0].dtype == torch.float32 and sd[k0].device.type == "cpu" and len(sd) == 256
    te = torch.load(os.path.join(path, "text_encoder_lora_EMA.pth"))
    assert "text_model.encoder.layers.0.self_attn.q_proj.lora_linear_layer.down.weight" in te
    step = ck.load_state(b, path)
    assert step == 20 and b.opt_step == 1 and b.lr_step == 2 and b.ema[0].optimization_step == 8
    for x, y in ((a.unet.lora_bank, b.unet.lora_bank), (a.te.lora_bank, b.te.lora_bank)):
        for n in ("flat", "ema", "exp_avg", "exp_avg_sq"):
            assert torch.equal(getattr(x, n), getattr(y, n)), n
    assert torch.equal(torch.rand(3), expect_next)              # RNG stream continues where the checkpoint left it
    assert b.unet.refreshed == 1 and b.te.refreshed == 1
    # export = the four LoRA files only
    out, files = ck.export_checkpoint(path)
    assert out.endswith("checkpoint_tmp-20_exported") and sorted(files) == sorted(f for k in ("unet", "text_encoder") for f in ck.BANK_FILES[k])
    with pytest.raises(ValueError):
        ck.export_checkpoint(os.path.join(d, "nope"))
    # files written by the reference (no trainer_state) load by key; wrong shapes are rejected
    c = _fake_trainer(3)
    ck.load_lora_files(ck.trainer_banks(c), out)
    assert torch.equal(c.unet.lora_bank.flat, a.unet.lora_bank.flat) and torch.equal(c.te.lora_bank.ema, a.te.lora_bank.ema)
    bad = dict(sd)
    bad[k0] = torch.zeros(5, 5)
    torch.save(bad, os.path.join(out, "unet_lora.pth"))
    with pytest.raises(ValueError):
        ck.load_lora_files(ck.trainer_banks(c), out)
    # rolling clean-up (:120-137): before the save at most limit-1 stay, oldest removed first, other names untouched
    for s in (40, 60, 100):
        os.makedirs(os.path.join(d, f"checkpoint_tmp-{s}"))
    os.makedirs(os.path.join(d, "checkpoint-200"))
    removed = ck.clean_checkpoint(d, "checkpoint_tmp", 2)
    assert removed == ["checkpoint_tmp-20", "checkpoint_tmp-40", "checkpoint_tmp-60"]
    assert sorted(os.listdir(d)) == ["checkpoint-200", "checkpoint_tmp-100", "checkpoint_tmp-20_exported"]


def test_pretrained_directory_layout_and_legacy_names(tmp_path):
    """A diffusers-layout directory (safetensors and .bin, legacy VAE attention names, prefix-less CLIP keys) loads into
    exactly the tensors the modules consume; missing tensors and wrong shapes are loud."""
    from safetensors.torch import save_file
    from finetune_fair_diffusion_amd import pretrained as P, weights as W
    from util_models import TINY_UNET, TINY_VAE, TINY_CLIP
    TINY_UNET, TINY_VAE, TINY_CLIP = W.UNetConfig(**TINY_UNET), W.VAEConfig(**TINY_VAE), W.CLIPTextConfig(**TINY_CLIP)
    m = tmp_path / "sd"
    for sub in ("unet", "vae", "text_encoder"):
        (m / sub).mkdir(parents=True)
    u = W.synthetic_state_dict(W.unet_param_shapes(TINY_UNET), seed=1)
    save_file({k: v.contiguous() for k, v in u.items()}, str(m / "unet" / "diffusion_pytorch_model.safetensors"))
    v = W.synthetic_state_dict(W.vae_param_shapes(TINY_VAE), seed=2)
    legacy = {}
    for k, t in v.items():
        for new, old in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            if f"attentions.0.{new}." in k:
                k = k.replace(f"attentions.0.{new}.", f"attentions.0.{old}.")
        legacy[k] = t
    legacy["encoder.conv_in.weight"] = torch.zeros(3)            # encoder half is ignored
    torch.save(legacy, str(m / "vae" / "diffusion_pytorch_model.bin"))
    c = W.synthetic_state_dict(W.clip_param_shapes(TINY_CLIP), seed=3)
    save_file({k[len("text_model."):]: t.contiguous() for k, t in c.items()}, str(m / "text_encoder" / "model.safetensors"))
    gu, gv, gc = P.load_unet(str(m), TINY_UNET), P.load_vae(str(m), TINY_VAE), P.load_text_encoder(str(m), TINY_CLIP)
    for got, ref in ((gu, u), (gv, v), (gc, c)):
        assert set(got) == set(ref)
        assert all(torch.equal(got[k], ref[k].float()) for k in ref)
    clf = W.synthetic_state_dict(W.mobilenet_param_shapes(6), seed=4)
    torch.save({"state_dict": {"model." + k: t for k, t in clf.items()}}, str(tmp_path / "clf.pt"))
    gl = P.load_classifier(str(tmp_path / "clf.pt"), 6)
    assert all(torch.equal(gl[k], clf[k].float()) for k in clf)
    with pytest.raises(ValueError):
        P.load_classifier(str(tmp_path / "clf.pt"), 80)          # 6-logit head into an 80-logit model
    del legacy["post_quant_conv.weight"]
    torch.save(legacy, str(m / "vae" / "diffusion_pytorch_model.bin"))
    with pytest.raises(KeyError):
        P.load_vae(str(m), TINY_VAE)
    with pytest.raises(FileNotFoundError):
        P.load_unet(str(tmp_path / "absent"), TINY_UNET)


def test_regulariser_asset_loaders(tmp_path, monkeypatch):
    """opensphere backbone saved from nn.DataParallel ('module.' prefix), face_feats.pkl triple, a CLIP directory that also holds
    the text tower, a DINOv2 hub checkpoint with its unused mask_token; missing locations for the hub-fetched encoders are loud."""
    import pickle, types
    from safetensors.torch import save_file
    from finetune_fair_diffusion_amd import pretrained as P, weights as W
    fn = W.synthetic_state_dict(W.sfnet20_param_shapes(in_size=112), seed=5)
    torch.save({"module." + k: v for k, v in fn.items()}, str(tmp_path / "backbone.pth"))
    got = P.load_face_net(str(tmp_path / "backbone.pth"), 112)
    assert set(got) == set(fn) and all(torch.equal(got[k], fn[k].float()) for k in fn)
    feats = torch.randn(50, 512, generator=torch.Generator().manual_seed(1)) * 3
    with open(tmp_path / "face_feats.pkl", "wb") as f:
        pickle.dump((feats, torch.zeros(50), torch.zeros(50, 2)), f)
    db = P.load_face_db(str(tmp_path / "face_feats.pkl"))
    assert db.shape == (50, 512) and torch.allclose(db.norm(dim=-1), torch.ones(50), atol=1e-6)
    assert torch.allclose(db, torch.nn.functional.normalize(feats, dim=-1))
    ccfg = W.ViTConfig(kind="clip", image_size=28, patch_size=14, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64, projection_dim=16)
    dcfg = W.ViTConfig(kind="dino", image_size=28, patch_size=14, hidden_size=24, num_hidden_layers=2, num_attention_heads=2, intermediate_size=48, projection_dim=0,
                       layer_norm_eps=1e-6, pos_grid=3)
    cv, dn = W.synthetic_state_dict(W.vit_param_shapes(ccfg), seed=6), W.synthetic_state_dict(W.vit_param_shapes(dcfg), seed=7)
    (tmp_path / "clip").mkdir()
    save_file({**{k: v.contiguous() for k, v in cv.items()}, "text_model.embeddings.token_embedding.weight": torch.zeros(4, 4)}, str(tmp_path / "clip" / "model.safetensors"))
    torch.save({**dn, "mask_token": torch.zeros(1, 24)}, str(tmp_path / "dinov2_vitb14_pretrain.pth"))
    gc, gd = P.load_clip_vision(str(tmp_path / "clip"), ccfg), P.load_dino(str(tmp_path / "dinov2_vitb14_pretrain.pth"), dcfg)
    assert set(gc) == set(cv) and set(gd) == set(dn)
    args = types.SimpleNamespace(weight_loss_img=8.0, weight_loss_face=1.0, opensphere_model_path=str(tmp_path / "backbone.pth"),
                                 face_feats_path=str(tmp_path / "face_feats.pkl"), size_aligned_face=112)
    monkeypatch.delenv("FD_CLIP_VISION_DIR", raising=False)
    with pytest.raises(FileNotFoundError):
        P.load_regularisers(args, dict(clip_vision=ccfg, dino=dcfg))
    monkeypatch.setenv("FD_CLIP_VISION_DIR", str(tmp_path / "clip"))
    monkeypatch.setenv("FD_DINO_WEIGHTS", str(tmp_path / "dinov2_vitb14_pretrain.pth"))
    out = P.load_regularisers(args, dict(clip_vision=ccfg, dino=dcfg))
    assert set(out) == {"clip_vision", "dino", "face_net", "face_db"}


def test_hash_tokenizer_shapes_like_reference_calls():
    """Prompt: BOS, words, EOS, mask ones.  Uncond: BOS then EOS padding to the same length with mask [1,1,0...] (:1020-1026)."""
    from finetune_fair_diffusion_amd.train import HashTokenizer, load_prompts
    import types
    ids, m, uids, um = HashTokenizer()("a photo of the face of a doctor, a person")
    L = len(ids)
    assert L == 13 and ids[0] == 49406 and ids[-1] == 49407 and m.tolist() == [1] * L
    assert uids.tolist() == [49406] + [49407] * (L - 1) and um.tolist() == [1, 1] + [0] * (L - 2)
    assert all(320 <= int(i) < 40000 for i in ids[1:-1])
    assert HashTokenizer()("a photo of the face of a doctor, a person")[0].tolist() == ids.tolist()
    ps = load_prompts(types.SimpleNamespace(prompt_occupation_path="/nonexistent.json", synthetic=True))
    assert len(ps) == 12 and "doctor" in ps[0]
    with pytest.raises(FileNotFoundError):
        load_prompts(types.SimpleNamespace(prompt_occupation_path="/nonexistent.json", synthetic=False))


# ------------------------------------------------------------------------------------------ image encoders of the regularisers
def test_oracle_clip_vision_pinned_against_transformers():
    """oracle.nn_vit.CLIPVisionModelWithProjection == the installed transformers implementation on the same random weights
    (get_clip_feat, 1-main-debias.py:1139-1156 uses transformers' model)."""
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    from oracle import nn_vit as V
    c = V.ViTConfig(image_size=56, patch_size=14, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                    projection_dim=32, pos_grid=4)
    hc = CLIPVisionConfig(hidden_size=64, intermediate_size=128, projection_dim=32, num_hidden_layers=2, num_attention_heads=4, image_size=56,
                          patch_size=14, hidden_act="gelu")
    torch.manual_seed(0)
    hf = CLIPVisionModelWithProjection(hc).eval()
    sd = {k: v for k, v in hf.state_dict().items() if "position_ids" not in k}
    from finetune_fair_diffusion_amd import weights as W
    shapes = W.vit_param_shapes(W.ViTConfig(**c.__dict__))
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v) for k, v in shapes.items()}      # key names + shapes are transformers'
    mine = V.build(c, sd)
    x = torch.randn(3, 3, 56, 56)
    with torch.no_grad():
        a, b = hf(pixel_values=x).image_embeds, mine(x)
    assert float((a - b).abs().max()) < 1e-5 * float(a.abs().max())


def test_oracle_dinov2_pinned_against_transformers_dinov2():
    """oracle.nn_vit.DinoVisionTransformer (dinov2 hub names: fused qkv, ls1 / ls2 LayerScale, final-norm CLS token; get_dino_feat, 1-main-debias.py:1158-1175)
    == the installed transformers ``Dinov2Model`` -- an independent port of facebookresearch/dinov2 -- on the same random weights.  Without position
    interpolation (stored grid == image grid) the two must agree to fp32 rounding; with the 37 -> 16 interpolation of the real checkpoint the hub code the
    reference fetched scales by (g + 0.1) / M where newer ports pass the target size: the difference between the two is measured and bounded, not hidden."""
    from transformers import Dinov2Config, Dinov2Model
    from oracle import nn_vit as V

    def pair(pos_grid):
        c = V.ViTConfig(kind="dino", image_size=56, patch_size=14, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256,
                        projection_dim=0, layer_norm_eps=1e-6, pos_grid=pos_grid)
        hc = Dinov2Config(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, mlp_ratio=4, image_size=14 * pos_grid, patch_size=14,
                          layer_norm_eps=1e-6, hidden_act="gelu", layerscale_value=1.0, qkv_bias=True, use_swiglu_ffn=False)
        torch.manual_seed(0)
        hf = Dinov2Model(hc).eval()
        with torch.no_grad():
            for p_ in hf.parameters():
                p_.normal_(0.0, 0.05)
        h = hf.state_dict()
        sd = {"cls_token": h["embeddings.cls_token"], "pos_embed": h["embeddings.position_embeddings"],
              "patch_embed.proj.weight": h["embeddings.patch_embeddings.projection.weight"], "patch_embed.proj.bias": h["embeddings.patch_embeddings.projection.bias"],
              "norm.weight": h["layernorm.weight"], "norm.bias": h["layernorm.bias"]}
        for i in range(2):
            a, o = f"encoder.layer.{i}.", f"blocks.{i}."
            for n in ("norm1", "norm2"):
                sd[o + n + ".weight"], sd[o + n + ".bias"] = h[a + n + ".weight"], h[a + n + ".bias"]
            for wb in ("weight", "bias"):
                sd[o + "attn.qkv." + wb] = torch.cat([h[a + f"attention.attention.{t}.{wb}"] for t in ("query", "key", "value")], 0)
                sd[o + "attn.proj." + wb] = h[a + "attention.output.dense." + wb]
                sd[o + "mlp.fc1." + wb], sd[o + "mlp.fc2." + wb] = h[a + "mlp.fc1." + wb], h[a + "mlp.fc2." + wb]
            sd[o + "ls1.gamma"], sd[o + "ls2.gamma"] = h[a + "layer_scale1.lambda1"], h[a + "layer_scale2.lambda1"]
        return hf, V.build(c, sd)

    x = torch.randn(3, 3, 56, 56)
    hf, mine = pair(4)                                   # 56 / 14 = 4: stored grid == image grid, no interpolation
    with torch.no_grad():
        a, b = hf(pixel_values=x).last_hidden_state[:, 0], mine(x)
    assert float((a - b).abs().max()) < 1e-5 * float(a.abs().max())
    hf, mine = pair(9)                                   # stored 9 x 9, image 4 x 4: the interpolated table
    with torch.no_grad():
        a, b = hf(pixel_values=x, interpolate_pos_encoding=True).last_hidden_state[:, 0] if "interpolate_pos_encoding" in Dinov2Model.forward.__code__.co_varnames \
            else hf(pixel_values=x).last_hidden_state[:, 0], mine(x)
    rel = float((a - b).abs().max() / a.abs().max())
    print(f"dinov2 with interpolated position table: oracle (hub formula, scale (g + 0.1) / M) vs transformers {rel:.2e}")
    assert rel < 5e-2


def test_vit_inventories_and_feature_loss_gradient():
    from finetune_fair_diffusion_amd import weights as W
    from finetune_fair_diffusion_amd.vit import feature_loss_and_grad, _interpolate_pos
    from oracle import nn_vit as V
    n = lambda c: sum(int(np.prod(s)) for s in W.vit_param_shapes(c).values())  # noqa: E731
    assert n(W.CLIP_VIT_H14) == 632_076_800 and n(W.DINOV2_VITB14) == 86_579_712      # ViT-H/14 with projection; dinov2_vitb14 (without mask_token)
    # oracle modules accept the product inventories (names + shapes), both kinds
    for cfg in (W.ViTConfig(kind="dino", image_size=56, hidden_size=64, num_hidden_layers=1, num_attention_heads=2, intermediate_size=128,
                            projection_dim=0, layer_norm_eps=1e-6, pos_grid=6),
                W.ViTConfig(kind="clip", image_size=56, hidden_size=64, num_hidden_layers=1, num_attention_heads=2, intermediate_size=128,
                            projection_dim=16, pos_grid=4)):
        V.build(V.ViTConfig(**cfg.__dict__), W.synthetic_state_dict(W.vit_param_shapes(cfg), seed=1))
    # position-table interpolation: product host code == oracle
    pe = torch.randn(1, 1 + 36, 8)
    assert torch.allclose(_interpolate_pos(pe, 4), V.interpolate_pos_encoding(pe, 4), atol=1e-6)
    assert torch.equal(_interpolate_pos(pe, 6), pe)
    # 1 - cos loss and its gradient vs autograd
    g = torch.Generator().manual_seed(0)
    e = torch.randn(5, 12, generator=g, requires_grad=True)
    t = torch.nn.functional.normalize(torch.randn(5, 12, generator=g), dim=-1)
    w = torch.rand(5, generator=g)
    loss_ref = 1 - (torch.nn.functional.normalize(e, dim=-1) * t).sum(-1)
    (loss_ref * w).sum().backward()
    loss, de = feature_loss_and_grad(e.detach(), t, w)
    assert torch.allclose(loss, loss_ref.detach(), atol=1e-6) and torch.allclose(de, e.grad, atol=1e-6)


# ------------------------------------------------------------------------------------------ face-realism term pieces
def test_oracle_sfnet20_pinned_against_opensphere_golden():
    """oracle.nn_sfnet.SFNet20 reproduces the outputs of the reference's own opensphere sfnet20 (tests/golden/reference_sfnet20.json,
    generated by importing /root/reference/opensphere) on the same seeded weights and input; key names are the reference's."""
    from finetune_fair_diffusion_amd import weights as W
    from oracle import nn_sfnet as OS
    gold = json.load(open(os.path.join(HERE, "golden", "reference_sfnet20.json")))
    sd = W.synthetic_state_dict(W.sfnet20_param_shapes(), seed=gold["weights_seed"])
    assert list(sd.keys()) == gold["keys"] or set(sd.keys()) == set(gold["keys"])
    assert sum(v.numel() for v in sd.values()) == gold["n_params"] == 24_500_992
    net = OS.SFNet20().eval()
    net.load_state_dict(sd, strict=True)
    x = torch.rand(2, 3, 112, 112, generator=torch.Generator().manual_seed(gold["input_seed"])) * 2 - 1
    with torch.no_grad():
        y, y2 = net(x), net(torch.flip(x, [3]))
        f = OS.get_face_feats(net, x, normalize=False)
    ref, ref2 = torch.tensor(gold["out"]), torch.tensor(gold["out_flipped"])
    assert float((y - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    assert float((y2 - ref2).abs().max()) < 1e-4 * float(ref2.abs().max())
    assert torch.allclose(f, ref + ref2, atol=1e-3)


def test_face_alignment_host_math():
    """Umeyama similarity: exact recovery of a known transform, product == oracle restatement; the folded 2x3 sampling matrix equals
    the oracle's kornia-style normalise / affine_grid / grid_sample chain (bilinear taps evaluated on the CPU)."""
    from finetune_fair_diffusion_amd.fairness import ALIGNED_FACE_LANDMARKS, SyntheticFaceProvider, alignment_sampling_matrix, umeyama_similarity
    from oracle import nn_sfnet as OS
    th, sc, t = 0.3, 1.7, np.array([5.0, -3.0])
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    src = np.random.RandomState(0).rand(5, 2) * 100
    T = umeyama_similarity(src, (sc * (R @ src.T)).T + t)
    assert np.abs(T[:2, :2] - sc * R).max() < 1e-12 and np.abs(T[:2, 2] - t).max() < 1e-10
    assert np.array_equal(ALIGNED_FACE_LANDMARKS, OS.SRC_LANDMARKS)
    H = W = 64
    crop = 28
    rng = np.random.RandomState(1)
    lm = OS.SRC_LANDMARKS / 112 * 30 + np.array([17.0, 12.0]) + rng.randn(5, 2)
    assert np.allclose(umeyama_similarity(lm, OS.SRC_LANDMARKS), OS.umeyama(lm, OS.SRC_LANDMARKS))
    A = alignment_sampling_matrix(lm, H, W, crop).reshape(2, 3)
    img = torch.rand(3, H, W, generator=torch.Generator().manual_seed(0)) * 2 - 1
    ref = OS.image_pipeline(img, lm, crop).numpy()
    ys, xs = np.meshgrid(np.arange(crop), np.arange(crop), indexing="ij")
    px, py = A[0, 0] * xs + A[0, 1] * ys + A[0, 2], A[1, 0] * xs + A[1, 1] * ys + A[1, 2]
    x0, y0 = np.floor(px).astype(int), np.floor(py).astype(int)
    lx, ly = px - x0, py - y0

    def at(c, yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = np.full(yy.shape, -1.0)
        v[ok] = img[c].numpy()[yy[ok], xx[ok]]
        return v
    out = np.stack([(1 - ly) * ((1 - lx) * at(c, y0, x0) + lx * at(c, y0, x0 + 1)) + ly * ((1 - lx) * at(c, y0 + 1, x0) + lx * at(c, y0 + 1, x0 + 1))
                    for c in range(3)])
    assert np.abs(out - ref).max() < 1e-4
    # the synthetic provider's landmarks sit inside its raw detector box and map back to the template under the estimated transform
    lms = SyntheticFaceProvider().landmarks(torch.zeros(2, 3, 512, 512))
    assert lms.shape == (2, 5, 2) and float(lms.min()) > 128 and float(lms.max()) < 384
    M = umeyama_similarity(lms[0].numpy(), ALIGNED_FACE_LANDMARKS)
    back = (M[:2, :2] @ lms[0].numpy().T).T + M[:2, 2]
    assert np.abs(back - ALIGNED_FACE_LANDMARKS).max() < 1e-3


def test_multi_attribute_regulariser_rules_golden():
    """exp-3 / exp-4 gen_dynamic_weights and apply_grad_hook_face (reference outputs) vs the product's attribute-count-generic forms."""
    from finetune_fair_diffusion_amd.fairness import face_grad_factors_multi, gen_dynamic_weights_multi
    for exp in ("exp3", "exp4"):
        for c in GOLD[f"gen_dynamic_weights_{exp}"]:
            w = gen_dynamic_weights_multi(torch.tensor(c["face_indicators"]), [torch.tensor(t) for t in c["targets"]],
                                          [torch.tensor(p) for p in c["preds_ori"]], c["factors"])
            assert np.allclose(w.numpy(), c["weights"], atol=1e-6), (exp, w, c["weights"])
        for c in GOLD[f"apply_grad_hook_face_{exp}"]:
            H = len(c["grad_ratio_ch0"][0])
            rects, facs = face_grad_factors_multi(torch.tensor(c["bbox"]), torch.tensor(c["bbox_ori"]), [torch.tensor(t) for t in c["targets"]],
                                                  [torch.tensor(p) for p in c["preds_ori"]], c["factors"], H, H)
            mask = torch.ones(len(c["bbox"]), H, H)
            for i, (r, f) in enumerate(zip(rects.tolist(), facs.tolist())):
                mask[i, r[1]:r[3], r[0]:r[2]] = f
            assert torch.allclose(mask, torch.tensor(c["grad_ratio_ch0"]), atol=1e-4), exp


def test_exp5_prompt_mix(tmp_path):
    """exp-5 :934-947: occupation prompts + 6x / 20x / 4x of the three extra prompt files."""
    import types
    from finetune_fair_diffusion_amd.train import load_prompts
    occ = tmp_path / "occ.json"
    occ.write_text(json.dumps({"prompt_templates_train": ["a {occupation}", "the {occupation}"], "occupations_train_set": ["x", "y", "z"]}))
    files = {}
    for k, n in (("prompt_occupation_w_style_and_context_path", 2), ("prompt_personal_descroptor_path", 1), ("prompt_sports_path", 3)):
        f = tmp_path / (k + ".json")
        f.write_text(json.dumps({"train_prompts": [f"{k}-{i}" for i in range(n)]}))
        files[k] = str(f)
    ps = load_prompts(types.SimpleNamespace(prompt_occupation_path=str(occ), synthetic=False, **files))
    assert len(ps) == 6 + 2 * 6 + 1 * 20 + 3 * 4 and ps[:2] == ["a x", "a y"] and ps[6] == "prompt_occupation_w_style_and_context_path-0"
    assert len(load_prompts(types.SimpleNamespace(prompt_occupation_path=str(occ), synthetic=False))) == 6
    files["prompt_sports_path"] = str(tmp_path / "missing.json")
    with pytest.raises(FileNotFoundError):
        load_prompts(types.SimpleNamespace(prompt_occupation_path=str(occ), synthetic=False, **files))


def test_oracle_clip_text_pinned_against_transformers():
    """oracle.nn_clip.CLIPTextModel == the installed transformers CLIPTextModel on the same random weights, called the way the
    reference calls it (1-main-debias.py:1011-1014, :1078-1081, :1087-1099): causal mask PLUS the tokenizer's attention_mask, for the
    prompt (all ones) and for the uncond "" prompt padded to L with mask [1,1,0,...,0]; hidden_act quick_gelu; output [0] =
    last_hidden_state.  (transformers 5.x drops the ``text_model.`` key prefix of 4.30 -- the oracle keeps the 4.30 names.)"""
    from transformers import CLIPTextConfig as HFConfig, CLIPTextModel as HFModel
    from oracle import nn_clip as C
    kw = dict(vocab_size=1000, hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4, max_position_embeddings=77)
    torch.manual_seed(0)
    hf = HFModel(HFConfig(hidden_act="quick_gelu", layer_norm_eps=1e-5, bos_token_id=998, eos_token_id=999, pad_token_id=999,
                          attn_implementation="eager", **kw)).eval()
    sd = {}
    for k, v in hf.state_dict().items():
        if "position_ids" in k:
            continue
        sd[k if k.startswith("text_model.") else "text_model." + k] = v
    mine = C.CLIPTextModel(C.CLIPTextConfig(**kw)).eval()
    missing, unexpected = mine.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    # and the product's inventory carries exactly these (4.30) names and shapes
    from finetune_fair_diffusion_amd import weights as W
    shapes = W.clip_param_shapes(W.CLIPTextConfig(**{k: v for k, v in kw.items() if k != "max_position_embeddings"}))
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v) for k, v in shapes.items()}
    L = 9
    ids = torch.tensor([[998, 5, 17, 400, 23, 8, 77, 300, 999], [998] + [999] * (L - 1)])
    mask = torch.tensor([[1] * L, [1, 1] + [0] * (L - 2)])
    with torch.no_grad():
        a = hf(input_ids=ids, attention_mask=mask)[0]
        b = mine(ids, mask)[0]
        a1 = hf(input_ids=ids[:1], attention_mask=mask[:1])[0]
    assert a.shape == b.shape == (2, L, 64)
    # rows whose query position attends to at least one unmasked key are defined identically; in the uncond sequence every query
    # row i >= 0 sees key 0 (BOS) through the causal mask, so all rows are comparable
    assert float((a - b).abs().max()) < 1e-5 * float(a.abs().max()), float((a - b).abs().max())
    assert float((a1 - b[:1]).abs().max()) < 1e-5 * float(a.abs().max())
    # the padding mask matters: without it the uncond sequence's rows 2.. change
    with torch.no_grad():
        c = mine(ids, None)[0]
    assert float((c[1, 2:] - b[1, 2:]).abs().max()) > 1e-3 and float((c[0] - b[0]).abs().max()) < 1e-6


# ------------------------------------------------------------------ raw accelerator.save_state directories (VERDICT r2 missing 6)
class _ProcLayers(torch.nn.Module):
    """Stand-in for diffusers 0.19.3 ``AttnProcsLayers`` (1-main-debias.py:818): a ModuleList of LoRA attention processors whose state-dict
    keys are renamed from ``layers.<i>`` to the processor names by a state-dict hook."""

    def __init__(self, names, shapes):
        super().__init__()
        procs = []
        for p in names:
            m = torch.nn.Module()
            for w in ("to_q", "to_k", "to_v", "to_out"):
                lo = torch.nn.Module()
                lo.down = torch.nn.Linear(shapes[f"{p}.{w}_lora.down.weight"][1], shapes[f"{p}.{w}_lora.down.weight"][0], bias=False)
                lo.up = torch.nn.Linear(shapes[f"{p}.{w}_lora.up.weight"][1], shapes[f"{p}.{w}_lora.up.weight"][0], bias=False)
                setattr(m, w + "_lora", lo)
            procs.append(m)
        self.layers = torch.nn.ModuleList(procs)
        self.mapping = dict(enumerate(names))

        def map_to(module, state_dict, *a, **k):
            return {k_.replace(f"layers.{k_.split('.')[1]}", module.mapping[int(k_.split('.')[1])]): v for k_, v in state_dict.items()}
        self._register_state_dict_hook(map_to)


class _CustomModel(torch.nn.Module):
    """The reference's text-encoder LoRA container (:856-872)."""

    def __init__(self, d):
        super().__init__()
        self.params = torch.nn.ParameterList([d[n] for n in d])


class _EMA:
    """State-dict surface of diffusers ``EMAModel`` as accelerate's ``register_for_checkpointing`` uses it."""

    def __init__(self, params, step):
        self.shadow_params, self.optimization_step = [p.detach().clone() * 0.5 for p in params], step

    def state_dict(self):
        return dict(decay=0.996, min_decay=0.0, optimization_step=self.optimization_step, update_after_step=0, use_ema_warmup=False, inv_gamma=1.0,
                    power=2 / 3, shadow_params=self.shadow_params)

    def load_state_dict(self, sd):
        self.shadow_params = sd["shadow_params"]


def test_resume_from_a_raw_accelerate_save_state_directory(tmp_path):
    """A directory written by REAL ``accelerate`` for the reference's object layout (:1648-1657, :2050-2068: CustomModel for the text-encoder LoRA,
    AttnProcsLayers for the U-Net LoRA, AdamW over chain(unet, text-encoder) parameters, LambdaLR, two registered EMA models) restores by NAME
    into the product's flat parameter banks: weights, Adam moments + step, lr position, EMA shadows, RNG streams, global step."""
    import types
    from accelerate import Accelerator
    from finetune_fair_diffusion_amd import accelerate_state as AS, weights as W
    from finetune_fair_diffusion_amd.layers import ParamBank
    from finetune_fair_diffusion_amd.step import EMAState
    ucfg = W.UNetConfig(block_out_channels=(32, 64), attention_head_dim=2, cross_attention_dim=16, down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                        up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"))
    ccfg = W.CLIPTextConfig(vocab_size=100, hidden_size=16, intermediate_size=32, num_hidden_layers=2, num_attention_heads=2)
    ushapes, tshapes = W.unet_lora_param_shapes(ucfg, 4), W.clip_lora_param_shapes(ccfg, 4)
    un_order, te_order = AS.unet_reference_param_order(ucfg), AS.te_reference_param_order(2)
    assert set(un_order) == set(ushapes) and set(te_order) == set(tshapes)
    # the reference names the entries of the parameter LIST ``_modify_text_encoder`` returns by searching named_parameters() for each of them
    # (:836-842): the order is the list's (attention q, k, v, out of every layer, then fc1, fc2 of every layer), NOT named_parameters()'
    from oracle import nn_clip
    te_o = nn_clip.CLIPTextModel(nn_clip.CLIPTextConfig(vocab_size=100, hidden_size=16, intermediate_size=32, num_hidden_layers=2, num_attention_heads=2))
    lora_list = nn_clip.modify_text_encoder(te_o, 4)
    name_order = [next(n for n, q in te_o.named_parameters() if q is p) for p in lora_list]
    assert name_order == te_order and name_order != [n for n, _ in te_o.named_parameters() if "lora_linear_layer" in n]
    assert te_order[0].endswith("layers.0.self_attn.q_proj.lora_linear_layer.down.weight") and te_order[16].endswith("layers.0.mlp.fc1.lora_linear_layer.down.weight")
    g = torch.Generator().manual_seed(0)
    procs = []
    for n in un_order:
        p = n.rsplit(".", 3)[0]
        if p not in procs:
            procs.append(p)
    unet_layers = _ProcLayers(procs, ushapes)
    for p in unet_layers.parameters():
        p.data = torch.randn(p.shape, generator=g)
    assert list(unet_layers.state_dict().keys()) == un_order
    te_dict = {n: torch.nn.Parameter(torch.randn(tshapes[n], generator=g)) for n in te_order}
    te_model = _CustomModel(te_dict)
    acc = Accelerator(cpu=True)
    params = list(unet_layers.parameters()) + list(te_model.parameters())          # chain(unet, text encoder) (:891)
    opt = torch.optim.AdamW(params, lr=5e-5)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    opt, sched = acc.prepare(opt, sched)
    te_ema, un_ema = _EMA(list(te_model.parameters()), 3), _EMA(list(unet_layers.parameters()), 3)
    te_model = acc.prepare(te_model); acc.register_for_checkpointing(te_ema)
    unet_layers = acc.prepare(unet_layers); acc.register_for_checkpointing(un_ema)
    for _ in range(3):
        for p in params:
            p.grad = torch.randn(p.shape, generator=g)
        opt.step(); sched.step()
    torch.manual_seed(4242); expect_rand = None
    d = str(tmp_path / "checkpoint_tmp-30")
    acc.save_state(d)
    expect_rand = torch.rand(3)           # what the reference would draw next after resuming
    assert AS.is_accelerate_state_dir(d)
    # the product side: flat banks on the CPU behind the trainer surface the reader touches
    ub, tb = ParamBank(ushapes, torch.device("cpu")), ParamBank(tshapes, torch.device("cpu"))
    stub = lambda **k: types.SimpleNamespace(**k)  # noqa: E731
    tr = stub(args=stub(train_unet=True, train_text_encoder=True), prefix=None, rank=0,
              unet=stub(lora_bank=ub, config=ucfg, refresh_lora=lambda: None), te=stub(lora_bank=tb, config=ccfg, refresh_lora=lambda: None),
              banks=[ub, tb], ema=[EMAState(0.996), EMAState(0.996)], opt_step=0, lr_step=0)
    torch.manual_seed(1)
    step = AS.load_accelerate_state(tr, d)
    assert step == 30 and tr.opt_step == 3 and tr.lr_step == 3 and [e.optimization_step for e in tr.ema] == [3, 3]
    sd_u = unet_layers.state_dict()
    opt_sd = opt.state_dict()
    for i, n in enumerate(un_order):
        assert torch.equal(ub.view(n), sd_u[n]) and torch.equal(ub.view(n, ub.ema), un_ema.shadow_params[i])
        assert torch.equal(ub.view(n, ub.exp_avg), opt_sd["state"][i]["exp_avg"]) and torch.equal(ub.view(n, ub.exp_avg_sq), opt_sd["state"][i]["exp_avg_sq"])
    for i, n in enumerate(te_order):
        assert torch.equal(tb.view(n), te_dict[n].data) and torch.equal(tb.view(n, tb.ema), te_ema.shadow_params[i])
        assert torch.equal(tb.view(n, tb.exp_avg), opt_sd["state"][len(un_order) + i]["exp_avg"])
    assert torch.equal(torch.rand(3), expect_rand)          # the torch CPU stream continues where the checkpoint left it
    # a run that trains something else is refused, not silently mis-mapped
    tr2 = stub(args=stub(train_unet=True, train_text_encoder=False), prefix=None, rank=0, unet=tr.unet, te=stub(lora_bank=None, config=ccfg), banks=[ub],
               ema=[EMAState(0.996)], opt_step=0, lr_step=0)
    with pytest.raises(ValueError):
        AS.load_accelerate_state(tr2, d)
    # a checkpoint of a TWO-process reference run: AcceleratedScheduler advanced the LambdaLR twice per optimiser step (last_epoch = 2 x global
    # step) and wrote one random_states file per rank; this build's lr position counts one unit per step
    import shutil
    d2 = str(tmp_path / "checkpoint_tmp-3")
    shutil.copytree(d, d2)
    sch = torch.load(os.path.join(d2, "scheduler.bin"), weights_only=False)
    sch["last_epoch"] = 6
    torch.save(sch, os.path.join(d2, "scheduler.bin"))
    shutil.copy(os.path.join(d2, "random_states_0.pkl"), os.path.join(d2, "random_states_1.pkl"))
    tr.lr_step = 0
    assert AS.load_accelerate_state(tr, d2, restore_rng=False) == 3 and tr.lr_step == 3
    # a different LoRA rank is refused BEFORE anything is copied (the banks keep what they held)
    ub8 = ParamBank(W.unet_lora_param_shapes(ucfg, 8), torch.device("cpu"))
    ub8.flat.fill_(7.0)
    tr8 = stub(args=stub(train_unet=True, train_text_encoder=True), prefix=None, rank=0,
               unet=stub(lora_bank=ub8, config=ucfg, refresh_lora=lambda: None), te=tr.te, banks=[ub8, tb], ema=[EMAState(0.996), EMAState(0.996)], opt_step=0, lr_step=0)
    before = tb.flat.clone()
    with pytest.raises(ValueError):
        AS.load_accelerate_state(tr8, d, restore_rng=False)
    assert bool((ub8.flat == 7.0).all()) and torch.equal(tb.flat, before)
