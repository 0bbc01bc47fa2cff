#!/bin/bash
# round-3 GPU pass G: step-time jitter diagnosis (with / without the Python GC), tests touched by the LoRA-refresh and prefetch changes, bench.
set -x
O=gpurun_out/r03g
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scratch/diag_step_jitter.py > $O/jitter.txt 2>&1
cat $O/jitter.txt | cut -c1-330
DIAG_NOGC=1 timeout 600 python scratch/diag_step_jitter.py > $O/jitter_nogc.txt 2>&1
grep "^step" $O/jitter_nogc.txt
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -6 > $O/pytest.log
cat $O/pytest.log
timeout 900 python bench.py --steps 8 --warmup 2 --no_cpu_baseline > $O/bench.json 2> $O/bench.err
python -c "import sys,json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'], d['config']['host_ms_per_step'])"
