#!/bin/bash
# round-3 GPU pass O: dQ kernel with the key mask behind a wave-uniform branch (product library) and attention built without SLP
# vectorisation (bench-hooks library, -fno-slp-vectorize: the guide prices v_pk_*_f32 beside MFMAs as an anti-lever); parity; step A/B.
set -x
O=gpurun_out/r03o
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" 2>&1 | tail -3 > $O/pytest_attn.log
cat $O/pytest_attn.log
timeout 600 python scratch/mb_attn_tr.py > $O/mb_attn_base.txt 2>&1
grep "^B" $O/mb_attn_base.txt | cut -c1-250
export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
timeout 600 python scratch/mb_attn_tr.py > $O/mb_attn_noslp.txt 2>&1
grep "^B" $O/mb_attn_noslp.txt | cut -c1-250
one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"; }
for m in noslp base noslp base; do
  if [ $m = noslp ]; then export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so; else unset FAIRDIFF_LIB; fi
  timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "ATTN=$m" | tee -a $O/step_ab.txt
done
