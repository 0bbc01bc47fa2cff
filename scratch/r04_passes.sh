#!/bin/bash
# every gpurun pass of round 4 is one case branch:  bash scratch/r04_passes.sh <letter>
mkdir -p gpurun_out
P=$PWD/finetune_fair_diffusion_amd
B="python bench.py --no_cpu_baseline --no_roofline"
case "$1" in
a)  # the fixed library (no packed-fp32 VALU, multi-row LayerNorm backward shipped): race detectors, then whole-step A/B against the round-3 library
    timeout 300 python scratch/diag_hazard.py s3x3 delayx4 > gpurun_out/r04a_hazard_fixed_lib.txt 2>&1
    timeout 600 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "delay_injection or reproduce_themselves or shipped_schedule" > gpurun_out/r04a_race_tests.txt 2>&1
    for i in 1 2; do
      $B --steps 6 --warmup 2 > gpurun_out/r04a_bench_new_$i.json 2> gpurun_out/r04a_bench_new_$i.err
      FAIRDIFF_LIB=$P/libfairdiff_hip_slp_r03.so $B --steps 6 --warmup 2 > gpurun_out/r04a_bench_r03lib_$i.json 2> gpurun_out/r04a_bench_r03lib_$i.err
    done
    grep -h "differ" gpurun_out/r04a_hazard_fixed_lib.txt | sort | uniq -c | head; tail -5 gpurun_out/r04a_race_tests.txt
    python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04a_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],3), round(d['ms_per_step'],1), d['config']['phase_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
    ;;
b)  # full GPU suite + smoke on the library without packed-fp32 VALU (multi-row LayerNorm backward, gather warp backward)
    timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r04b_gpu_suite.log 2>&1; tail -25 gpurun_out/r04b_gpu_suite.log
    timeout 300 python __graft_entry__.py smoke > gpurun_out/r04b_smoke.log 2>&1; tail -2 gpurun_out/r04b_smoke.log
    ;;
c)  # which change moved the exp-3 tiny-model gradient cosine (0.981 in round 2 -> 0.966)?  same test under the A/B switches
    T="python -m pytest tests/test_engine_gpu.py -x -q -s -k test_full_step_multi_attribute_exp3"
    (echo "## default"; $T; echo "## FD_HOST_SCALES=1"; FD_HOST_SCALES=1 $T; echo "## round-3 library"; FAIRDIFF_LIB=$P/libfairdiff_hip_slp_r03.so $T; echo "## round-3 library + host scales"; FD_HOST_SCALES=1 FAIRDIFF_LIB=$P/libfairdiff_hip_slp_r03.so $T) 2>&1 | grep -i "##\|cosine\|passed\|failed" > gpurun_out/r04c_exp3_cosine_ab.txt
    cat gpurun_out/r04c_exp3_cosine_ab.txt
    ;;
*) echo "unknown pass $1";;
esac
