#!/bin/bash
# round-2 GPU pass K: tile-policy thresholds, second sweep (bench-hooks library)
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export FAIRDIFF_LIB=$R/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
run() {
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/r02_pol_$name.json 2> gpurun_out/r02_pol_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/r02_pol_$name.json')); print('$name', '$*', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms')"
}
run k_ref FD_GEMM_T256=100 FD_GEMM_T128=80
run k_64_48 FD_GEMM_T256=64 FD_GEMM_T128=48
run k_32_32 FD_GEMM_T256=32 FD_GEMM_T128=32
run k_100_80_t160_80 FD_GEMM_T256=100 FD_GEMM_T128=80 FD_GEMM_T160=80
run k_64_48_t160_80 FD_GEMM_T256=64 FD_GEMM_T128=48 FD_GEMM_T160=80
run k_64_64_t160_100_vae100 FD_GEMM_T256=64 FD_GEMM_T128=64 FD_GEMM_T160=100 FD_GEMM_TVAE=100
run k_100_48 FD_GEMM_T256=100 FD_GEMM_T128=48
run k_base A=1
