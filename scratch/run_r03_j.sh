#!/bin/bash
# round-3 GPU pass J: the new device OT solver test + persistent streaming GEMM: correctness (bit-equal to the shipped kernels) and isolated A/B.
set -x
O=gpurun_out/r03j
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "ot_assign" 2>&1 | tail -5 > $O/pytest_ot.log
cat $O/pytest_ot.log
export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
FD_GEMM_PPS_MIN=1 timeout 600 python scratch/mb_pps.py > $O/mb_pps_min1.txt 2>&1
cat $O/mb_pps_min1.txt | cut -c1-200
FD_GEMM_PPS_MIN=1 FD_GEMM_PPS_WG=512 timeout 600 python scratch/mb_pps.py > $O/mb_pps_wg512.txt 2>&1
tail -32 $O/mb_pps_wg512.txt | cut -c1-200
