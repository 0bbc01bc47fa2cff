#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests/ -q -m gpu --durations=12 ) > gpurun_out/r02_gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r02_gpu_suite.log
grep -E "passed|failed|rc=|^E |real|s call|s setup" gpurun_out/r02_gpu_suite.log | cut -c1-200 | tail -25
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r02_bench_i.json 2> gpurun_out/r02_bench_i.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_i.json')); print(round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['traffic'], d['cpu_baseline']['value'])"
