#!/bin/bash
mkdir -p gpurun_out
for v in 3 4 2 3 4; do
FD_BWD_STREAMS=$v timeout 900 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/r02_bench_y.json 2> gpurun_out/r02_bench_y.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r02_bench_y.json').read().strip().splitlines()[-1])
print('bwd_streams=$v', round(d['value'],3),'img/s',round(d['ms_per_step'],1),'ms bwd', d['config']['phase_ms']['R3_bwd_unet'], 'rollouts', d['config']['phase_ms']['R1_rollout'])
PY
done
