"""BASELINE configs[4]'s working dtype (bf16; its e4m3 attention left the product in round 5, see scratch/attn_fp8_experiment.hip): the checks live in tests/run_bf16_checks.py and run in their own
process, because a process's working dtype -- which library it loads and what ``ops.F16`` is -- is fixed at import (FD_DTYPE)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(extra_env, which):
    env = dict(os.environ, FD_DTYPE="bf16", **extra_env)
    env.pop("FAIRDIFF_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "run_bf16_checks.py")] + which, env=env, capture_output=True, text=True, timeout=1500)
    print(r.stdout[-6000:])
    print(r.stderr[-3000:])
    assert r.returncode == 0 and "BF16 CHECKS PASSED" in r.stdout


def test_bf16_kernels_unet_and_training_step(dev):
    _run({}, ["kernels", "tiny", "sd15"])


def test_exp3_step_in_bf16(dev):
    """BASELINE configs[4]'s working dtype on a multi-attribute step: exp-3 logic on the bf16 library."""
    _run({}, ["exp3"])
