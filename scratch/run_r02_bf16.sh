#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py --dtype bf16 --no_cpu_baseline > gpurun_out/r02_bench_bf16.json 2> gpurun_out/r02_bench_bf16.err; echo rc=$?
timeout 900 python bench.py --dtype bf16 --fp8_attn --no_cpu_baseline > gpurun_out/r02_bench_bf16_fp8.json 2> gpurun_out/r02_bench_bf16_fp8.err; echo rc=$?
python - <<'PY'
import json
for f in ("gpurun_out/r02_bench_bf16.json", "gpurun_out/r02_bench_bf16_fp8.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value'],3),'img/s',round(d['ms_per_step'],1),'ms', d['dtype'], d['config']['workload'][:60])
PY
