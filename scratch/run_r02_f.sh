#!/bin/bash
# round-2 GPU pass F: main-loop ablation of the big-tile kernels; two-stream backward A/B + parity
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
echo "== ablation"
timeout 600 python scratch/mb_ablate.py > gpurun_out/r02_ablate.txt 2>&1; cat gpurun_out/r02_ablate.txt | grep -v amdgpu.ids
echo "== step tests with the two-stream backward"
timeout 1200 python -m pytest tests/test_engine_gpu.py -q -x -k "full_fairness_step or shared_mode or multi_attribute_with_oracle" > gpurun_out/r02_f_engine.log 2>&1; echo "rc=$?" >> gpurun_out/r02_f_engine.log
grep -E "passed|failed|rc=|^E " gpurun_out/r02_f_engine.log | cut -c1-300 | tail -8
echo "== bench default"
timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/r02_bench_f.json 2> gpurun_out/r02_bench_f.err; echo "rc=$?"; tail -2 gpurun_out/r02_bench_f.err | cut -c1-300
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_f.json')); print('default:', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms')"
FD_NO_CONCURRENT_BWD=1 timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/r02_bench_f_nobwd.json 2> gpurun_out/r02_bench_f_nobwd.err
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_f_nobwd.json')); print('FD_NO_CONCURRENT_BWD:', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms')"
