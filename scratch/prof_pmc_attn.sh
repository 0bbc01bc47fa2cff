#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc_attn -o a -- python3 $R/scratch/mb_attn40.py > $R/gpurun_out/pmc_attn.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_attn2 -o a -- python3 $R/scratch/mb_attn40.py > $R/gpurun_out/pmc_attn2.log 2>&1
tail -2 $R/gpurun_out/pmc_attn.log | cut -c1-200; tail -2 $R/gpurun_out/pmc_attn2.log | cut -c1-200
