#!/bin/bash
# round-3 GPU pass K: whole-step A/B of the persistent streaming GEMM policy (bench-hooks library, same box).
set -x
O=gpurun_out/r03k
mkdir -p $O
export TMPDIR=/tmp
export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"; }
run() { FD_GEMM_PP=$1 FD_GEMM_PPS_MIN=$2 FD_GEMM_PPS_MAXK=$3 timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "PP=$1 PPS_MIN=$2 MAXK=$3" | tee -a $O/step_ab.txt; }
run 45 512 100000
run 173 1 100000
run 173 256 100000
run 173 512 100000
run 45 512 100000
run 173 1 700
run 173 257 100000
