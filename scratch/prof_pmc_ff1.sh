#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export FD_GEMM_BIGK=320
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc_ff1 -o a -- python3 $R/scratch/mb_ff1.py > $R/gpurun_out/pmc_ff1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_ff1b -o a -- python3 $R/scratch/mb_ff1.py > $R/gpurun_out/pmc_ff1b.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_ff1c -o a -- python3 $R/scratch/mb_ff1.py > $R/gpurun_out/pmc_ff1c.log 2>&1
tail -2 $R/gpurun_out/pmc_ff1c.log | cut -c1-200
