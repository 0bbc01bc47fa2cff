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
*) echo "unknown pass $1";;
esac
