#!/bin/bash
# full GPU suite + smoke + default bench after the attention occupancy / fused GroupNorm / exp-2 / detector-provider changes
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu --durations=8 ) > gpurun_out/r02_gpu_suite_w.log 2>&1; echo rc=$?; tail -14 gpurun_out/r02_gpu_suite_w.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2; echo rc=$?
echo skip-bench
cat <<PY > /dev/null
import json
d=json.loads(open('gpurun_out/r02_bench_w.json').read().strip().splitlines()[-1])
print(round(d['value'],3),'img/s',round(d['ms_per_step'],1),'ms', d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['traffic'], d['cpu_baseline']['value'])
print(d['config']['phase_ms'])
PY
