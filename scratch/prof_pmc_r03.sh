#!/bin/bash
# separate --pmc passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2: they do not fit one pass), kernel-trace only (no other trace domains)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_r03_fetch -o f -- python3 $R/scratch/mb_pmc_r03.py > $R/gpurun_out/pmc_r03_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_r03_write -o w -- python3 $R/scratch/mb_pmc_r03.py > $R/gpurun_out/pmc_r03_write.log 2>&1
cd $R
python3 scratch/pmc_r03_summary.py
