#!/bin/bash
# LDS-side counters of gemm_big_kernel<256,320,conv> (is the LDS array the shared resource of fragment reads and direct-to-LDS writes?)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_lds_$tag -o p -- python3 $R/scratch/mb_pmc_conv.py > $R/gpurun_out/pmc_lds_$tag.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_lds_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "gemm_big_kernel" in r["Kernel_Name"]:
            e = agg[r["Counter_Name"]]; e[0] += float(r["Counter_Value"]); e[1] += 1
    for k, (v, n) in agg.items():
        print(f"{k:40s} per launch {v / n:16.0f}   ({n} launches)")
PY
