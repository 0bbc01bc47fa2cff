#!/bin/bash
# round-2 GPU pass J: tile-policy A/B of the WHOLE step under the multi-stream schedule (bench-hooks library, environment switches)
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
export FAIRDIFF_LIB=$R/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
run() {
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/r02_pol_$name.json 2> gpurun_out/r02_pol_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/r02_pol_$name.json')); print('$name', '$*', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms')"
}
run base A=1
run maxsplit1 FD_GEMM_MAXSPLIT=1
run maxsplit2 FD_GEMM_MAXSPLIT=2
run maxsplit4 FD_GEMM_MAXSPLIT=4
run thr_lo FD_GEMM_T256=100 FD_GEMM_T128=80
run thr_mid FD_GEMM_T256=128 FD_GEMM_T128=100
run thr_hi FD_GEMM_T256=400 FD_GEMM_T128=300
run base2 A=1
