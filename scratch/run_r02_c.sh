#!/bin/bash
# round-2 GPU pass C: per-shape GEMM table + rocprof kernel trace of one (shared-mode) bench step
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 900 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --dump_shapes gpurun_out/r02_gemm_shapes.csv > gpurun_out/r02_bench_c.json 2> gpurun_out/r02_bench_c.err
echo "rc=$?"; head -60 gpurun_out/r02_gemm_shapes.csv
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02_a -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/gpurun_out/prof_r02_a.log 2>&1
cd $R
DB=$(find gpurun_out/prof_r02_a -name "*.db" | head -1)
python scratch/profsum.py $DB gpurun_out/r02_kernel_stats_a.csv 45
