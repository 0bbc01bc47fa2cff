#!/bin/bash
# round-3 GPU pass H: lagged gradient scales (no mid-step host syncs) -- tests and same-box A/B against FD_SYNC_SCALES=1.
set -x
O=gpurun_out/r03h
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -q -x 2>&1 | tail -6 > $O/pytest_engine.log
cat $O/pytest_engine.log
timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_two_rank_gpu.py -m gpu -q -x -k "shipped or mixed or golden or two_rank" 2>&1 | tail -6 > $O/pytest_full.log
cat $O/pytest_full.log
one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'], d['config']['host_ms_per_step'])"; }
for m in sync lagged sync lagged; do
  if [ $m = sync ]; then export FD_SYNC_SCALES=1; else unset FD_SYNC_SCALES; fi
  timeout 600 python bench.py --steps 8 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "SCALES=$m" | tee -a $O/step_ab.txt
done
