#!/bin/bash
# final validation of the round: full GPU suite, smoke, default bench (with CPU baseline), kernel trace of one step
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu --durations=6 ) > gpurun_out/r02_gpu_suite_x.log 2>&1; echo rc=$?; tail -12 gpurun_out/r02_gpu_suite_x.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; echo rc=$?
timeout 1200 python bench.py > gpurun_out/r02_bench_x.json 2> gpurun_out/r02_bench_x.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_bench_x.json').read().strip().splitlines()[-1])
print(round(d['value'],3),'img/s',round(d['ms_per_step'],1),'ms', d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['traffic'], d['cpu_baseline']['value'])
print(d['config']['phase_ms'])
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02_x -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/gpurun_out/prof_r02_x.log 2>&1
cd $R
DB=$(find gpurun_out/prof_r02_x -name "*.db" | head -1)
python scratch/profsum.py $DB gpurun_out/r02_kernel_stats_x.csv 16
