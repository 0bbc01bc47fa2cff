#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_engine_gpu.py -q -x -k "full_fairness_step or shared_mode or multi_attribute_with_oracle or smooth_head or train_loop" > gpurun_out/r02_p_engine.log 2>&1; echo "rc=$?" >> gpurun_out/r02_p_engine.log
grep -E "passed|failed|rc=|^E " gpurun_out/r02_p_engine.log | cut -c1-300 | tail -6
timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > gpurun_out/r02_bench_p.json 2> gpurun_out/r02_bench_p.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_p.json')); print(round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['traffic'])"
