#!/bin/bash
# round-3 GPU pass F: R2 prefetch under the tail -- equivalence test, whole-step A/B over the number of prefetched denoising steps.
set -x
O=gpurun_out/r03f
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "prefetch or full_fairness_step or exp2 or train_loop" 2>&1 | tail -15 > $O/pytest.log
cat $O/pytest.log
one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['r2_steps_prefetched_under_previous_tail'], d['config']['phase_ms'], d['config']['host_ms_per_step'])"; }
for k in 0 4 6 8 10 0 6; do
  FD_R2_PREFETCH_STEPS=$k timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>$O/err_$k.txt | one "R2_PREFETCH=$k" | tee -a $O/step_ab.txt
done
