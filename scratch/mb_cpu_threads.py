"""Thread scaling of the CPU oracle on the GPU box's host ("use os.cpu_count() threads, or state why 64 is faster, with a
measurement"): one SD-v1.5-size U-Net CFG-pair forward (no grad) at 32 / 64 / 128 / 256 torch threads -> profiles/r03_cpu_baseline_thread_scaling.txt"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util_models as U  # noqa: E402

om = U.oracle_models(rank=4, train_unet=False, train_te=True, lora_up_std=0.01, size="sd15", eval_copies=False)
x, enc = torch.randn(2, 4, 64, 64), torch.randn(2, 13, 768)
print("logical CPUs:", os.cpu_count())
for th in (8, 16, 24, 32, 48, 64):     # 128 -> 9.6 s, 256 -> 150 s (first run of this probe, profiles/r03_cpu_baseline_thread_scaling.txt)
    torch.set_num_threads(th)
    with torch.no_grad():
        om["unet"](x, torch.tensor(500), encoder_hidden_states=enc)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            om["unet"](x, torch.tensor(500), encoder_hidden_states=enc)
            ts.append(time.perf_counter() - t0)
    print(f"{th:4d} threads: U-Net CFG-pair forward {min(ts):.2f} s (min of 3: {[round(t, 2) for t in ts]})  -> {2 * 0.798 / min(ts):.3f} TFLOP/s")
