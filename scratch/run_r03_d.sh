#!/bin/bash
# round-3 GPU pass D: the complete -m gpu suite with durations (budget: < 900 s), backward-stream and tile-threshold A/Bs of the whole step
# now that the ping-pong kernels carry the convolutions, the default bench line, a kernel trace.
set -x
O=gpurun_out/r03d
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
BL=$R/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
( time timeout 2400 python -m pytest tests -m gpu -q -x --durations=30 ) > $O/pytest_all.log 2>&1
tail -50 $O/pytest_all.log
one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"; }
for st in 2 3 4; do
  FD_BWD_STREAMS=$st timeout 600 python bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "BWD_STREAMS=$st" | tee -a $O/step_ab.txt
done
for th in "100 80" "64 48" "140 110" "200 160"; do
  set -- $th
  FAIRDIFF_LIB=$BL FD_GEMM_T256=$1 FD_GEMM_T128=$2 timeout 600 python bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "T256=$1,T128=$2" | tee -a $O/step_ab.txt
done
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 1200 $O/bench.json
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o r03d -- python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
cd $R
DB=$(find /tmp/prof_d -name "*.db" | head -1)
python scratch/profsum.py $DB $O/kernel_stats.csv 40 > $O/kernel_stats_top.txt
cat $O/kernel_stats_top.txt
timeout 600 python scratch/mb_cpu_threads.py > $O/cpu_threads.txt 2>&1
cat $O/cpu_threads.txt
