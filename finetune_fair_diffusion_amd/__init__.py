"""MI355X-native fairness-finetuning hot path (sail-sg/finetune-fair-diffusion, exp-1/3/4/5 ``1-main-debias.py:1746-2029``).

The 16-bit working dtype of a process is fixed when the package is first imported (``lib.WORKING_DTYPE``): ``FD_DTYPE=fp16|bf16``
in the environment, or -- so that the reference's own flag keeps working from the command line -- ``--mixed_precision bf16`` /
a ``--config`` YAML carrying ``mixed_precision: bf16`` in ``sys.argv`` (exp-1-debias-gender/1-main-debias.py:401-405, :625-642).
"""
import os
import sys


def _preselect_working_dtype(argv):
    if "FD_DTYPE" in os.environ:
        return
    mp = None
    for i, a in enumerate(argv):
        if a == "--config" and i + 1 < len(argv) and os.path.exists(argv[i + 1]):
            try:
                import yaml
                with open(argv[i + 1]) as f:
                    mp = (yaml.safe_load(f) or {}).get("mixed_precision", mp)
            except Exception:
                pass
    for i, a in enumerate(argv):
        if a == "--mixed_precision" and i + 1 < len(argv):
            mp = argv[i + 1]
        elif a.startswith("--mixed_precision="):
            mp = a.split("=", 1)[1]
    if mp == "bf16":
        os.environ["FD_DTYPE"] = "bf16"


_preselect_working_dtype(sys.argv)
