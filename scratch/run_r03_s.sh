#!/bin/bash
# round-3 GPU pass S: with 8 hardware queues (now the package default), re-tune the stream knobs that were tuned under 4: backward streams, R2 prefetch depth.
set -x
O=gpurun_out/r03s
mkdir -p $O
export TMPDIR=/tmp
one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['hip_hw_queues'], d['config']['phase_ms'])"; }
run() { env $1 timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "$1" | tee -a $O/step_ab.txt; }
run FD_X=0
run FD_BWD_STREAMS=4
run FD_R2_PREFETCH_STEPS=12
run FD_R2_PREFETCH_STEPS=16
run FD_X=0
run FD_BWD_STREAMS=2
run FD_R2_PREFETCH_STEPS=20
run GPU_MAX_HW_QUEUES=12
