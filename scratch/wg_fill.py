"""How full is the chip under the SHIPPED multi-stream schedule?    rocprofv3's kernel trace serialises most dispatches, so the
measurement is in-kernel: the bench-hooks library's instrumented kernels (common.h FD_WG_TRACE: every GEMM / convolution / attention / norm / GEGLU / LoRA-wgrad
kernel, > 95 % of the step's kernel time) log (kernel id, waves, CU, start, end) per WORKGROUP on the chip-wide 100 MHz clock while bench.py's own loop runs.
From the last three steps: share of CU-time with at least one workgroup resident (per step and per phase), resident workgroups / waves per CU, the distribution
of the number of busy CUs over time, and CU-time per kernel family.
usage: python scratch/wg_fill.py [out.txt] [steps]"""
import ctypes, json, os, sys, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FAIRDIFF_LIB"] = os.path.join(ROOT, "finetune_fair_diffusion_amd", "libfairdiff_hip_bench.so")
sys.path.insert(0, ROOT)
out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "wg_fill.txt")
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
import finetune_fair_diffusion_amd  # noqa: F401
import numpy as np
import torch
from finetune_fair_diffusion_amd import lib, step as step_mod

NAMES = {1: "gemm_glds", 2: "gemm_big", 3: "splitk_reduce", 4: "gemm_skinny", 5: "gemm_pp", 6: "conv_halo", 7: "attn_fwd", 8: "attn_bwd_dq", 9: "attn_bwd_dkdv", 10: "cross_block",
         11: "gn_reduce", 12: "gn_apply", 13: "gn_bwd_apply", 14: "gn_fused_fwd", 15: "gn_fused_bwd", 16: "layernorm", 17: "geglu_bwd", 18: "lora_wgrad_partial", 19: "lora_wgrad_final", 20: "add"}
CAP = 110_000_000                     # records (32 bytes each: 3.5 GB)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
buf = torch.zeros(4 + 4 * CAP, dtype=torch.int64, device=dev)
L = lib.get()
L.fd_bench_wg_trace.argtypes = [ctypes.c_void_p, ctypes.c_uint]
L.fd_bench_wg_trace.restype = None
L.fd_bench_wg_trace(buf.data_ptr(), CAP)
marks = []
orig = step_mod.FairnessTrainer.train_step


def traced(self, *a, **k):
    if len(marks) < 2:                             # the two warm-up steps: restart the log (stream-ordered), only the timed steps are kept
        buf[0:1].zero_()
    m = torch.empty(1, dtype=torch.int64).pin_memory()
    m.copy_(buf[0:1], non_blocking=True)         # records written when the launch stream reaches this point
    marks.append(m)
    return orig(self, *a, **k)


step_mod.FairnessTrainer.train_step = traced
import bench
sys.argv = ["bench.py", "--steps", str(nsteps), "--warmup", "2", "--no_cpu_baseline", "--no_roofline"]
so = io.StringIO()
with contextlib.redirect_stdout(so):
    bench.main()
torch.cuda.synchronize()
line = json.loads(so.getvalue().strip().splitlines()[-1])
L.fd_bench_wg_trace(None, 0)
n = int(buf[0].item())
assert n <= CAP, f"trace overflow: {n} records"
rec = buf[4:4 + 4 * n].view(n, 4).cpu().numpy().astype(np.uint64)
starts = [int(m.item()) for m in marks] + [n]
phase_ms = line["config"]["phase_ms"]
with open(out_path, "w") as f:
    def P(*a):
        print(*a)
        print(*a, file=f)
    P(f"workgroup trace of bench.py --steps {nsteps} --warmup 2 (bench-hooks library, shipped schedule: two rollout streams, three backward streams, R2 prefetch)")
    P(f"bench line of the traced run: {line['value']:.3f} images/s, {line['ms_per_step']:.1f} ms per step, phases {phase_ms}")
    P(f"{n} workgroup records in all; instrumented kernels: {', '.join(NAMES.values())}")
    for s in range(len(starts) - 1 - nsteps, len(starts) - 1):
        r = rec[starts[s]:starts[s + 1]]
        if len(r) == 0:
            continue
        kid = (r[:, 0] & np.uint64(0xff)).astype(np.int64)
        waves = ((r[:, 0] >> np.uint64(8)) & np.uint64(0xff)).astype(np.int64)
        cu = (((r[:, 1] >> np.uint64(32)) & np.uint64(0xf)) << np.uint64(8) | ((r[:, 1] >> np.uint64(8)) & np.uint64(0xff))).astype(np.int64)
        t0, t1 = r[:, 2].astype(np.int64), r[:, 3].astype(np.int64)
        T0, T1 = int(t0.min()), int(t1.max())
        wall = (T1 - T0) / 1e5                      # ms (100 MHz ticks)
        cus = np.unique(cu)
        ncu = len(cus)
        # sweep per CU: busy time (>= 1 workgroup resident), workgroup-time, wave-time; and the chip-wide count of busy CUs over time in 1 ms bins
        nb = int(np.ceil(wall)) + 1
        busy_bins = np.zeros(nb)
        busy_total = wg_time = wave_time = 0.0
        order = np.lexsort((t0, cu))
        cu_s, a_s, b_s, w_s = cu[order], t0[order], t1[order], waves[order]
        bounds = np.flatnonzero(np.diff(cu_s)) + 1
        for lo, hi in zip(np.r_[0, bounds], np.r_[bounds, len(cu_s)]):
            a, b = a_s[lo:hi], b_s[lo:hi]
            wg_time += float((b - a).sum())
            wave_time += float(((b - a) * w_s[lo:hi]).sum())
            end = np.maximum.accumulate(b)
            new = np.r_[True, a[1:] > end[:-1]]          # an interval that starts after everything before it has ended opens a busy span
            sa = a[new]
            sb = np.r_[end[:-1][new[1:]], end[-1]]
            busy_total += float((sb - sa).sum())
            for x, y in zip((sa - T0) / 1e5, (sb - T0) / 1e5):   # spans -> 1 ms bins
                i, j = int(x), int(y)
                if i == j:
                    busy_bins[i] += y - x
                else:
                    busy_bins[i] += i + 1 - x
                    busy_bins[i + 1:j] += 1.0
                    busy_bins[j] += y - j
        P(f"\n== step {s - (len(starts) - 1 - nsteps) + 1} of {nsteps}: {len(r)} workgroups on {ncu} CUs, {wall:.1f} ms from the first start to the last end (includes the prefetched R2 steps that overlap the neighbours)")
        P(f"   CU-time with >= 1 workgroup resident: {100 * busy_total / (ncu * (T1 - T0)):.1f} % of {ncu} CUs x wall;  resident workgroups per CU (time average) {wg_time / (ncu * (T1 - T0)):.2f};  resident waves per CU {wave_time / (ncu * (T1 - T0)):.2f} of 32")
        # phases by the bench's own marks (cumulative from the step's first record)
        edges, acc_ms = [], 0.0
        for k in ("R1_rollout", "R1_vae", "classify_targets", "R2_tail_and_regularisers", "R3_loss_and_image_grad", "R3_bwd_vae", "R3_bwd_unet", "sync_update"):
            edges.append((k, acc_ms, acc_ms + phase_ms.get(k, 0.0)))
            acc_ms += phase_ms.get(k, 0.0)
        for k, x, y in edges:
            i, j = int(x), min(int(np.ceil(y)), nb)
            if j > i and y - x >= 2.0:
                P(f"   phase {k:26s} [{x:7.1f}, {y:7.1f}) ms: busy CUs {busy_bins[i:j].mean():6.1f} of {ncu} ({100 * busy_bins[i:j].mean() / ncu:5.1f} %)")
        q = np.percentile(busy_bins[:int(wall)], [5, 25, 50, 75, 95])
        P(f"   busy CUs per 1 ms bin: p5 {q[0]:.0f}  p25 {q[1]:.0f}  median {q[2]:.0f}  p75 {q[3]:.0f}  p95 {q[4]:.0f};  bins below 128 busy CUs: {int((busy_bins[:int(wall)] < 128).sum())} of {int(wall)}")
        P("   busy CUs, 25 ms bins: " + " ".join(f"{busy_bins[i:i + 25].mean():.0f}" for i in range(0, int(wall), 25)))
        tot = float((t1 - t0).sum())
        P("   workgroup-time by kernel: " + ", ".join(f"{NAMES.get(k, k)} {100 * float((t1 - t0)[kid == k].sum()) / tot:.1f} %" for k in np.argsort(-np.bincount(kid, weights=(t1 - t0).astype(np.float64), minlength=21))[:12]))
