#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "groupnorm" > gpurun_out/r02_u_gn_tests.log 2>&1; echo rc=$?; tail -5 gpurun_out/r02_u_gn_tests.log
timeout 300 python scratch/mb_gn.py 2>&1 | tail -8
timeout 900 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > gpurun_out/r02_bench_u.json 2> gpurun_out/r02_bench_u.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_bench_u.json').read().strip().splitlines()[-1])
print(d['value'],'img/s',d['ms_per_step'],'ms', d['config']['phase_ms'])
print(d['config']['phase_ms_single_stream'])
PY
