#!/bin/bash
# kernel trace of one shipped step after the attention occupancy targets
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02_s -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/gpurun_out/prof_r02_s.log 2>&1
cd $R
DB=$(find gpurun_out/prof_r02_s -name "*.db" | head -1)
python scratch/profsum.py $DB gpurun_out/r02_kernel_stats_s.csv 30
