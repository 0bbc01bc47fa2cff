#!/bin/bash
# round-3 GPU pass C: ping-pong kernels incl. the 128x320 / split-K variants (microbench A/B), whole-step A/B of the dispatch policies through the
# bench-hooks library, the kernel + full-size suites with the shipped policy, the default bench line and a kernel trace.
set -x
O=gpurun_out/r03c
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
BL=$R/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
FAIRDIFF_LIB=$BL timeout 900 python scratch/mb_pp.py > $O/mb_pp.txt 2>&1
cat $O/mb_pp.txt
for mode in 0 37 45 61 63; do
  FAIRDIFF_LIB=$BL FD_GEMM_PP=$mode timeout 600 python bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('PP_MODE $mode', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])" | tee -a $O/step_ab.txt
done
timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest_kernels.log
cat $O/pytest_kernels.log
timeout 2400 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -150 > $O/pytest_full.log
grep -n "rel max err\|cosine\|passed\|failed\|kept\|Error" $O/pytest_full.log | cut -c1-230 | tail -60
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o r03c -- python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
cd $R
DB=$(find /tmp/prof_c -name "*.db" | head -1)
python scratch/profsum.py $DB $O/kernel_stats.csv 40 > $O/kernel_stats_top.txt
cat $O/kernel_stats_top.txt
