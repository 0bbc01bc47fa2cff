#!/bin/bash
# round-2 GPU pass H: the complete -m gpu suite (wall time budget), smoke(), then profiles: kernel trace + PMC traffic of the roofline kernel
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=15 ) > gpurun_out/r02_gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r02_gpu_suite.log
grep -E "passed|failed|rc=|^E |real|s call" gpurun_out/r02_gpu_suite.log | cut -c1-200 | tail -25
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
bash scratch/prof_pmc_r02.sh 2>&1 | tail -40
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02_b -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/gpurun_out/prof_r02_b.log 2>&1
cd $R
DB=$(find gpurun_out/prof_r02_b -name "*.db" | head -1)
python scratch/profsum.py $DB gpurun_out/r02_kernel_stats_b.csv 30
