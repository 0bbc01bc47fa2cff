#!/bin/bash
# round-2 GPU pass O: complete -m gpu suite, smoke(), default bench on the final product build
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests/ -q -m gpu --durations=8 ) > gpurun_out/r02_gpu_suite_final.log 2>&1; echo "rc=$?" >> gpurun_out/r02_gpu_suite_final.log
grep -E "passed|failed|rc=|^E |real|s call" gpurun_out/r02_gpu_suite_final.log | cut -c1-200 | tail -16
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 900 python bench.py > gpurun_out/r02_bench_o.json 2> gpurun_out/r02_bench_o.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_o.json')); print(round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['traffic'], round(d['cpu_baseline']['value'],4))"
