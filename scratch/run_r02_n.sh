#!/bin/bash
# round-2 GPU pass N: zero-page hoist (new bench-hooks lib) on the ablation shapes; four-stage p4 kernel A/B -- with the ROUND-1 thresholds, so that
# the isolated microbenchmarks select the same tiles as the earlier ablation did (the adopted 100/80 thresholds are a whole-step choice)
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export FD_GEMM_T256=200 FD_GEMM_T128=160
python scratch/mb_ablate.py 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_zp_hoist_dbg0.txt
python scratch/mb_mb.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_p4_ab.txt
