#!/bin/bash
R=$GRAFT_REPO_ROOT
python3 $R/scratch/mb_conv.py 2>&1 | grep -E "gemm|conv"
echo "--- no banding"; FD_GEMM_L2_KB=100000 python3 $R/scratch/mb_conv.py 2>&1 | grep -E "gemm|conv"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_dfetch2 -o f -- python3 $R/scratch/mb_pmc_dense.py > $R/gpurun_out/pmc_dfetch2.log 2>&1
