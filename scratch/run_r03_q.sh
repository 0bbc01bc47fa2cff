#!/bin/bash
# round-3 GPU pass Q: configs[4] lines after the transpose-read attention (bf16 vs bf16 + e4m3 self-attention), exp-4 step with the device
# OT solver vs the host solver, RCCL collectives at world size 1.
set -x
O=gpurun_out/r03q
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --dtype bf16 --fp8_attn > $O/bench_bf16_fp8.json 2> $O/bench_bf16_fp8.err
timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --dtype bf16 > $O/bench_bf16_b.json 2> $O/bench_bf16_b.err
timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --dtype bf16 --fp8_attn > $O/bench_bf16_fp8_b.json 2> $O/bench_bf16_fp8_b.err
timeout 600 python bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline --experiment exp-4 > $O/bench_exp4_device_ot.json 2> $O/bench_exp4_device_ot.err
FD_OT_HOST=1 timeout 600 python bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline --experiment exp-4 > $O/bench_exp4_host_ot.json 2> $O/bench_exp4_host_ot.err
timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --force_collectives > $O/bench_rccl_ws1.json 2> $O/bench_rccl_ws1.err
for f in $O/*.json; do python -c "import sys,json; d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); c=d['config']; print('$f', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', c.get('ot_solver'), c.get('ot_solve_ms'), c.get('ot_exposed_wait_ms'))"; done
