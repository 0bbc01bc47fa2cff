#!/bin/bash
# round-3 GPU pass R: number of HIP hardware queues (GPU_MAX_HW_QUEUES, default 4) vs the step's streams (launch, R2, two more backward
# streams, OT, RCCL's internal one): streams that share a hardware queue serialise.  Whole-step A/B, with and without collectives.
set -x
O=gpurun_out/r03r
mkdir -p $O
export TMPDIR=/tmp
one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"; }
for q in 4 8 4 8 2 16; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "HWQ=$q" | tee -a $O/step_ab.txt
done
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline --force_collectives 2>/dev/null | one "HWQ=$q collectives" | tee -a $O/step_ab.txt
done
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "single_image_step or multi_attribute_with_oracle_ot" 2>&1 | tail -4 > $O/pytest.log
cat $O/pytest.log
