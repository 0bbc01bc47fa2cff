#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/r02_bench_v.json 2> gpurun_out/r02_bench_v.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_bench_v.json').read().strip().splitlines()[-1])
print(d['value'],'img/s',d['ms_per_step'],'ms')
print('device', d['config']['phase_ms'])
print('host  ', d['config']['host_enqueue_ms'])
PY
