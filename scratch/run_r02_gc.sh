#!/bin/bash
mkdir -p gpurun_out
for v in 0 1 0 1 0 1; do
FD_BENCH_GC=$v timeout 900 python bench.py --steps 6 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/r02_bench_gc.json 2> gpurun_out/r02_bench_gc.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r02_bench_gc.json').read().strip().splitlines()[-1])
print('gc_in_timed_region=$v', round(d['value'],3),'img/s',round(d['ms_per_step'],1),'ms  per-step host ms', d['config']['host_ms_per_step'])
PY
done
