#!/bin/bash
# round-3 GPU pass M: dK/dV transpose-read form at three waves per SIMD (bench-hooks library built with -DFD_DKDV_TR_W3) vs two; step A/B.
set -x
O=gpurun_out/r03m
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scratch/mb_attn_tr.py > $O/mb_attn_w2.txt 2>&1
grep "^B" $O/mb_attn_w2.txt | cut -c1-250
export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
timeout 600 python scratch/mb_attn_tr.py > $O/mb_attn_w3.txt 2>&1
grep "^B" $O/mb_attn_w3.txt | cut -c1-250
one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"; }
for m in w3 w2 w3 w2; do
  if [ $m = w3 ]; then export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so; else unset FAIRDIFF_LIB; fi
  timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "DKDV=$m" | tee -a $O/step_ab.txt
done
