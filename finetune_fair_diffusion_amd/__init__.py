"""MI355X-native fairness-finetuning hot path (sail-sg/finetune-fair-diffusion, exp-1/3/4/5 ``1-main-debias.py:1746-2029``).

The 16-bit working dtype of a process is fixed when the package is first imported (``lib.WORKING_DTYPE``): ``FD_DTYPE=fp16|bf16``
in the environment, or -- so that the reference's own flag keeps working from the command line -- ``--mixed_precision bf16`` /
a ``--config`` YAML carrying ``mixed_precision: bf16`` in ``sys.argv`` (exp-1-debias-gender/1-main-debias.py:401-405, :625-642).
"""
import os
import sys


def _preselect_working_dtype(argv):
    """Same precedence as ``cli.parse_args`` and the reference (:625-642): the YAML overlay is applied LAST, so a ``mixed_precision`` key
    in the ``--config`` file wins over the command-line flag.  Only the package's own entry points look at argv -- importing the package
    from a host program (pytest, a notebook) never parses that program's arguments."""
    if "FD_DTYPE" in os.environ:
        return
    mp = None
    for i, a in enumerate(argv):
        if a == "--mixed_precision" and i + 1 < len(argv):
            mp = argv[i + 1]
        elif a.startswith("--mixed_precision="):
            mp = a.split("=", 1)[1]
    for i, a in enumerate(argv):
        path = argv[i + 1] if (a == "--config" and i + 1 < len(argv)) else (a.split("=", 1)[1] if a.startswith("--config=") else None)
        if path and os.path.exists(path):
            try:
                import yaml
                with open(path) as f:
                    mp = (yaml.safe_load(f) or {}).get("mixed_precision", mp)
            except Exception:
                pass
    if mp == "bf16":
        os.environ["FD_DTYPE"] = "bf16"


def _is_package_entry_point(argv):
    """True when the process was started as ``python -m finetune_fair_diffusion_amd.train|generate ...`` (runpy leaves "-m" in argv[0] while
    the package is being imported) or as one of those files directly."""
    if not argv:
        return False
    if argv[0] == "-m":
        try:
            orig = sys.orig_argv
        except AttributeError:      # Python < 3.10
            return True
        import re
        for i, tok in enumerate(orig[1:-1], 1):     # "-m", or combined short flags ending in m: -um, -Im, -Bum ...
            if re.fullmatch(r"-[A-Za-z]*m", tok):
                return orig[i + 1].split(".")[0].replace("-", "_") == __name__.split(".")[0]
            if not tok.startswith("-"):
                break
        return False
    base = os.path.basename(argv[0])
    return base in ("train.py", "generate.py") and os.path.dirname(os.path.abspath(argv[0])) == os.path.dirname(os.path.abspath(__file__))


# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share one serialise.  The step keeps up to seven
# streams busy (launch, frozen-model rollout, two more backward streams, the OT solver's, RCCL's internal one, torch's copy streams): with 4
# queues the R2 prefetch and -- at world size > 1 -- every collective queued behind another stream's kernels (same box, bench.py: 1427-1431 ms
# with 4 queues, 1382-1399 with 8, 1383 with 16; with the step's collectives on: 1614 vs 1421 ms, profiles/r03_step_ab_hw_queues.txt).  The
# runtime reads the variable when it initialises the device, so it must be in the environment before the first HIP call; an explicit setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

if _is_package_entry_point(sys.argv):
    _preselect_working_dtype(sys.argv)
