#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention or attn" > gpurun_out/r02_r_attn_tests.log 2>&1; echo rc=$?; tail -3 gpurun_out/r02_r_attn_tests.log
timeout 900 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > gpurun_out/r02_bench_r.json 2> gpurun_out/r02_bench_r.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_bench_r.json').read().strip().splitlines()[-1])
print(d['value'],'img/s',d['ms_per_step'],'ms', d['config']['phase_ms'])
PY
