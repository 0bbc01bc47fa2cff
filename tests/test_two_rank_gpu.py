"""Two processes, one GPU, the PRODUCT's real ``train_step`` (reference: exp-1-debias-gender/1-main-debias.py:1805-1837
gather -> global targets, :1998-2011 gradient sync).  The multi-rank path hid a hang in round 2 that only a manual run found."""
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import util_models as U  # noqa: E402
import run_two_rank_step as R  # noqa: E402

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(experiment, out_dir, world=2, b_per_rank=None):
    env = dict(os.environ)
    for k in ("FD_DTYPE", "FAIRDIFF_LIB", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if b_per_rank is not None:
        env["FD_TEST_B_PER_RANK"] = str(b_per_rank)
    if world > 2:
        env["GPU_MAX_HW_QUEUES"] = "2"       # eight processes on ONE device: 8 x 8 hardware queues would oversubscribe it (a real run has one device per rank)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "run_two_rank_step.py"), experiment, str(out_dir)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    # no retry (round 4 retried the eight-rank launch once, "seen to fail once in five runs": a retry hides exactly the start-up race an eight-GPU node would
    # hit without one).  Ten consecutive launches of this test, first attempts only: profiles/r05_eight_rank_launch_10x.txt.
    print(r.stdout[-3000:])
    print(r.stderr[-3000:])
    for k in range(world):
        ep = os.path.join(out_dir, f"rank{k}.err")
        if os.path.exists(ep):
            print(f"---- rank {k} traceback\n" + open(ep).read()[-2500:])
    assert r.returncode == 0, f"{world}-rank run failed"
    return [torch.load(os.path.join(out_dir, f"rank{k}.pt")) for k in range(world)]


def test_two_ranks_equal_one_rank_with_twice_the_batch_exp1(dev, tmp_path):
    """exp-1, LoRA on U-Net AND text encoder: two ranks of 3 images (one micro-batch each) against ONE rank with all 6 images in two
    micro-batches of 3 -- the reference's own recipe for trading GPUs against per-GPU batch (exp-1 README:19-20).  The ranks must end the
    step with identical synced gradients and identical parameters; the pair must match the single process (same global targets; gradient
    to the fp16 noise of running the networks at batch 3 vs 6, which select different GEMM tiles)."""
    r0, r1 = _launch("exp-1", tmp_path)
    assert r0["finite"] and r1["finite"]
    for a, b in zip(r0["grads"] + r0["params"], r1["grads"] + r1["params"]):
        assert torch.equal(a, b)                                   # after the all-reduce every rank holds the same bits
    tr = R.build("exp-1", dev, 0, 1)
    out = tr.train_step(U.tiny_tokens(), R.global_noises(2), R.S)
    one = R.snapshot(tr, out)
    tg = torch.cat([r0["targets"]["gender"], r1["targets"]["gender"]])
    assert tg.tolist() == one["targets"]["gender"].tolist() and int((tg != -1).sum()) >= 3, (tg, one["targets"])
    lf = torch.cat([r0["loss_fair"], r1["loss_fair"]])
    assert float((lf - one["loss_fair"]).abs().max()) < 2e-2
    for k, (g2, g1) in enumerate(zip(r0["grads"], one["grads"])):
        cos = float(F.cosine_similarity(g2.double(), g1.double(), dim=0))
        ratio = float(g2.norm() / g1.norm())
        print(f"bank {k}: two ranks vs one rank with 2x batch: cosine {cos:.6f}  norm ratio {ratio:.4f}")
        assert cos > 0.995 and 0.97 < ratio < 1.03
    for p2, p1 in zip(r0["params"], one["params"]):
        assert float((p2 - p1).abs().max()) <= 2.5e-4               # one AdamW step of lr 5e-5: |delta| <= ~1e-4 per parameter


def test_two_ranks_multi_attribute_exchange_points_exp3(dev, tmp_path):
    """exp-3 (gender x race): the probability all-gather for both attributes, the OT-plan all-reduce and the gradient all-reduce on two
    ranks: both ranks must derive their targets from the same global plan (rank k's slice), take the same finite/non-finite branch and
    end with identical gradients and parameters."""
    r0, r1 = _launch("exp-3", tmp_path)
    assert r0["finite"] and r1["finite"]
    assert set(r0["targets"]) == {"gender", "race"}
    for a, b in zip(r0["grads"] + r0["params"], r1["grads"] + r1["params"]):
        assert torch.equal(a, b)
    assert float(r0["grads"][0].abs().max()) > 0
    n_t = sum(int((t != -1).sum()) for r in (r0, r1) for t in r["targets"].values())
    print("exp-3 two ranks: targets", {k: v.tolist() for k, v in r0["targets"].items()}, {k: v.tolist() for k, v in r1["targets"].items()})
    assert n_t >= 2


def test_eight_ranks_one_global_batch_exp1(dev, tmp_path, monkeypatch):
    """World size 8 through the product's ``train_step`` (rank-count assumptions -- this rank's slice of the global targets
    :1836, the gather order, the 1 / (world * N_backward) gradient scale :2005): eight processes (gloo, all on cuda:0; per-rank CPU shares from
    affinity.pin_rank are exercised by the launcher's LOCAL_WORLD_SIZE) with 2 images each against ONE process with all 16 images."""
    monkeypatch.setenv("FD_TEST_B_PER_RANK", "2")
    import importlib
    importlib.reload(R)
    try:
        ranks = _launch("exp-1", tmp_path, world=8, b_per_rank=2)
        assert all(r["finite"] for r in ranks)
        for r in ranks[1:]:
            for a, b in zip(ranks[0]["grads"] + ranks[0]["params"], r["grads"] + r["params"]):
                assert torch.equal(a, b)
        tr = R.build("exp-1", dev, 0, 1)
        out = tr.train_step(U.tiny_tokens(), R.global_noises(8), R.S)
        one = R.snapshot(tr, out)
        tg = torch.cat([r["targets"]["gender"] for r in ranks])
        assert tg.shape == (16,) and tg.tolist() == one["targets"]["gender"].tolist() and int((tg != -1).sum()) >= 6, (tg, one["targets"])
        for k, (g8, g1) in enumerate(zip(ranks[0]["grads"], one["grads"])):
            cos = float(F.cosine_similarity(g8.double(), g1.double(), dim=0))
            ratio = float(g8.norm() / g1.norm())
            print(f"bank {k}: eight ranks vs one rank with 8x batch: cosine {cos:.6f}  norm ratio {ratio:.4f}")
            assert cos > 0.995 and 0.97 < ratio < 1.03
    finally:
        monkeypatch.delenv("FD_TEST_B_PER_RANK")
        importlib.reload(R)


def test_eight_ranks_multi_attribute_exp4(dev, tmp_path, monkeypatch):
    """BASELINE configs[3] is exp-4 on EIGHT ranks: the 8-logit head (gender x race x age), the device OT solver, the all-gather of all three
    attributes' probabilities, the [n, 16] summed-plan all-reduce and the gradient all-reduce at world size 8 through the product's ``train_step`` -- eight
    processes with 2 images each (gloo, all on cuda:0).  Every rank must hold the same bits after the step and derive its slice of ONE global plan.  Against one
    process with all 16 images the Monte-Carlo plans differ by construction (every rank draws its own 100 capacity samples -- the reference's per-rank solve, exp-3
    :1488-1536 -- and the product sums the eight plans; one rank draws 100), so targets are compared as an agreement rate and the gradient by its cosine."""
    monkeypatch.setenv("FD_TEST_B_PER_RANK", "2")
    import importlib
    importlib.reload(R)
    try:
        ranks = _launch("exp-4", tmp_path, world=8, b_per_rank=2)
        assert all(r["finite"] for r in ranks)
        assert set(ranks[0]["targets"]) == {"gender", "race", "age"}
        for r in ranks[1:]:
            for a, b in zip(ranks[0]["grads"] + ranks[0]["params"], r["grads"] + r["params"]):
                assert torch.equal(a, b)
        assert float(ranks[0]["grads"][0].abs().max()) > 0
        tr = R.build("exp-4", dev, 0, 1)
        out = tr.train_step(U.tiny_tokens(), R.global_noises(8), R.S)
        one = R.snapshot(tr, out)
        # the probabilities every rank contributed are those of the single process (same images, same classifier): the exchange carried the right rows in rank order
        probs8 = torch.cat([r["probs"] for r in ranks])
        assert probs8.shape == one["probs"].shape and float((probs8 - one["probs"]).abs().max()) < 2e-2
        agree = total = nset = 0
        for a in ("gender", "race", "age"):
            t8 = torch.cat([r["targets"][a] for r in ranks])
            assert t8.shape == (16,) and t8.shape == one["targets"][a].shape
            both = (t8 != -1) & (one["targets"][a] != -1)
            agree += int((t8[both] == one["targets"][a][both]).sum())
            total += int(both.sum())
            nset += int((t8 != -1).sum())
            print(f"exp-4 eight ranks, {a}: targets {t8.tolist()} | one rank {one['targets'][a].tolist()}")
        g8, g1 = ranks[0]["grads"][0], one["grads"][0]
        cos = float(F.cosine_similarity(g8.double(), g1.double(), dim=0))
        print(f"exp-4 eight ranks vs one rank with 8x batch: {nset} targets set, agreement where both are set {agree} / {total}; gradient cosine {cos:.4f}  norm ratio {float(g8.norm() / g1.norm()):.3f}")
        assert nset >= 8 and total >= 6 and agree >= 0.6 * total
        assert cos > 0.5
    finally:
        monkeypatch.delenv("FD_TEST_B_PER_RANK")
        importlib.reload(R)

