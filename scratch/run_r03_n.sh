#!/bin/bash
# round-3 GPU pass N: dK/dV kernel with the second score tile's softmax interleaved with the first tile's dV / dK products (bench-hooks
# library built with -DFD_DKDV_PIPE) vs the shipped order; parity on the variant; step A/B.
set -x
O=gpurun_out/r03n
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scratch/mb_attn_tr.py > $O/mb_attn_base.txt 2>&1
grep "^B" $O/mb_attn_base.txt | cut -c1-250
export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
timeout 600 python scratch/mb_attn_tr.py > $O/mb_attn_pipe.txt 2>&1
grep "^B" $O/mb_attn_pipe.txt | cut -c1-250
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention_fwd_bwd" 2>&1 | tail -3 > $O/pytest_attn_pipe.log
cat $O/pytest_attn_pipe.log
one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"; }
for m in pipe base pipe base; do
  if [ $m = pipe ]; then export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so; else unset FAIRDIFF_LIB; fi
  timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "DKDV=$m" | tee -a $O/step_ab.txt
done
