#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_dfetch -o f -- python3 $R/scratch/mb_pmc_dense.py > $R/gpurun_out/pmc_dfetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_dwrite -o w -- python3 $R/scratch/mb_pmc_dense.py > $R/gpurun_out/pmc_dwrite.log 2>&1
ls $R/gpurun_out/pmc_dfetch $R/gpurun_out/pmc_dwrite
