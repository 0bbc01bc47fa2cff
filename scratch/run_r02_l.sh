#!/bin/bash
# round-2 GPU pass L: new tile thresholds + batched LoRA weight gradients: parity, then bench A/B
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x > gpurun_out/r02_l_kernels.log 2>&1; echo "rc=$?" >> gpurun_out/r02_l_kernels.log; tail -3 gpurun_out/r02_l_kernels.log | cut -c1-300
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py -q -x -k "unet_forward_and_backward or cfg_pair or full_fairness_step or shared_mode or sd15_unet_cfg or zero_init" > gpurun_out/r02_l_engine.log 2>&1; echo "rc=$?" >> gpurun_out/r02_l_engine.log
grep -E "passed|failed|rc=|^E " gpurun_out/r02_l_engine.log | cut -c1-300 | tail -8
for arm in default nobatch; do
  if [ $arm = nobatch ]; then export FD_NO_WGRAD_BATCH=1; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/r02_bench_l_$arm.json 2> gpurun_out/r02_bench_l_$arm.err
  python -c "
import json; d=json.load(open('gpurun_out/r02_bench_l_$arm.json')); print('$arm:', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"
done
