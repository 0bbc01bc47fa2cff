#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python3 $R/scratch/mb_pmc_conv.py > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o w -- python3 $R/scratch/mb_pmc_conv.py > $R/gpurun_out/pmc_write.log 2>&1
ls -la $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write; tail -3 $R/gpurun_out/pmc_fetch.log | cut -c1-200
