#!/bin/bash
# SQ counters (one pass, 8 slots) for the attention kernels (harness) and the GEMM / conv kernels (mb_pmc_r02.py): where do the wave-cycles go?
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_r02_sq_attn -o a -- $R/scratch/attn_fwd_experiment > $R/gpurun_out/pmc_r02_sq_attn.log 2>&1
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_r02_sq_gemm -o g -- python3 $R/scratch/mb_pmc_r02.py > $R/gpurun_out/pmc_r02_sq_gemm.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, os
for tag in ("attn", "gemm"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_r02_sq_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = open(f"gpurun_out/r02_pmc_sq_{tag}.txt", "w")
    hdr = f"{'kernel':72s} launches  WAVE_CYC   WAIT_ANY%  WAIT_INST%  ACTIVE%  MFMA_BUSY/BUSY_CYC  LDS_CONFLICT/LDS_ACTIVE"
    print(hdr); out.write(hdr + "\n")
    for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
        n = len(c.get("SQ_WAVE_CYCLES", []))
        if not n: continue
        m = {x: sum(v) / len(v) for x, v in c.items()}
        wc = m.get("SQ_WAVE_CYCLES", 1) or 1
        line = (f"{k:72s} {n:6d}  {wc:10.3e}  {100 * m.get('SQ_WAIT_ANY', 0) / wc:8.1f}  {100 * m.get('SQ_WAIT_INST_ANY', 0) / wc:9.1f}  "
                f"{100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc:7.1f}  {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(m.get('SQ_BUSY_CYCLES', 1), 1):12.3f}  "
                f"{m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1):12.3f}")
        print(line); out.write(line + "\n")
PY
