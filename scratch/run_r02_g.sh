#!/bin/bash
# round-2 GPU pass G: number of backward streams
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
for n in 2 3 4; do
  FD_BWD_STREAMS=$n timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/r02_bench_g$n.json 2> gpurun_out/r02_bench_g$n.err
  python -c "
import json; d=json.load(open('gpurun_out/r02_bench_g$n.json')); print('bwd streams $n:', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'], 'peak GB', d['config']['peak_hbm_gb'])"
done
