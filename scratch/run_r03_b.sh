#!/bin/bash
# round-3 GPU pass B: the two schedule-property tests after the tolerance split, the ping-pong GEMM A/B (bench-hooks library), GroupNorm
# microbenchmark, CPU-oracle thread scaling, and a rocprofv3 kernel trace of two bench steps summarised with scratch/profsum.py.
set -x
O=gpurun_out/r03b
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -k "shipped_schedule or mixed_keep or golden" 2>&1 | grep -v "^$" | tail -80 > $O/pytest_sched.log
grep -n "rel max err\|cosine\|passed\|failed\|kept" $O/pytest_sched.log | tail -30
FAIRDIFF_LIB=$R/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so timeout 900 python scratch/mb_pp.py > $O/mb_pp.txt 2>&1
cat $O/mb_pp.txt
timeout 600 python scratch/mb_gn.py > $O/mb_gn.txt 2>&1
tail -30 $O/mb_gn.txt
timeout 900 python scratch/mb_cpu_threads.py > $O/cpu_threads.txt 2>&1
cat $O/cpu_threads.txt
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o r03b -- python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
cd $R
DB=$(find /tmp/prof_b -name "*.db" | head -1)
python scratch/profsum.py $DB $O/kernel_stats.csv 45 > $O/kernel_stats_top.txt
cat $O/kernel_stats_top.txt
