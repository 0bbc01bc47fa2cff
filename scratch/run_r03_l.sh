#!/bin/bash
# round-3 GPU pass L: attention kernels with LDS transpose reads: parity (both forms, bit-identity), isolated A/B, whole-step A/B.
set -x
O=gpurun_out/r03l
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" 2>&1 | tail -8 > $O/pytest_attn.log
cat $O/pytest_attn.log
timeout 600 python scratch/mb_attn_tr.py > $O/mb_attn_tr.txt 2>&1
cat $O/mb_attn_tr.txt | cut -c1-250
one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"; }
for m in tr copies tr copies; do
  if [ $m = copies ]; then export FD_ATTN_NO_TR=1; else unset FD_ATTN_NO_TR; fi
  timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "ATTN=$m" | tee -a $O/step_ab.txt
done
