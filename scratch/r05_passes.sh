#!/bin/bash
# every gpurun pass of round 5 is one case branch:  bash scratch/r05_passes.sh <letter>
mkdir -p gpurun_out
P=$PWD/finetune_fair_diffusion_amd
R=$PWD
B="python bench.py --no_cpu_baseline --no_roofline"
bench_table() {   # bench_table <glob>: value, ms/step, median, phases of every bench line that matches
python - "$1" <<'PY'
import json, glob, statistics, sys
for f in sorted(glob.glob(sys.argv[1])):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); h = d['config']['host_ms_per_step']
        print(f, round(d['value'], 3), round(d['ms_per_step'], 1), 'median', round(statistics.median(h), 1), d['config']['phase_ms'])
    except Exception as e:
        print(f, 'ERR', e)
PY
}
case "$1" in
a)  # evidence first: per-dispatch kernel trace of the shipped three-stream step (grid sizes, concurrency timeline), per-shape single-stream
    # times (--dump_shapes), then the in-situ PMC passes: SQ wave-cycle breakdown, FETCH_SIZE, WRITE_SIZE (separate passes, kernel-trace only)
    O=gpurun_out/r05a; mkdir -p $O
    cd /tmp && export TMPDIR=/tmp
    timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_r05a -o r05a -- python3 $R/bench.py --steps 2 --warmup 2 --no_cpu_baseline --no_roofline > $R/$O/bench_trace.json 2> $R/$O/bench_trace.err
    cd $R
    DB=$(find /tmp/prof_r05a -name "*.db" | head -1)
    python scratch/profsum.py $DB $O/kernel_stats.csv 30 > $O/kernel_stats_top.txt
    python scratch/r05_trace_rows.py $DB $O > $O/trace_rows.txt 2>&1; tail -45 $O/trace_rows.txt | cut -c1-200
    timeout 900 python bench.py --steps 4 --warmup 2 --no_cpu_baseline --dump_shapes $O/gemm_shapes.csv > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
    KR='gemm|attn|gn_|layernorm|geglu|splitk'
    cd /tmp
    C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES"
    timeout 1200 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "$KR" --output-format csv -d /tmp/pmc_r05a_sq -o s -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/pmc_sq.log 2>&1
    timeout 1200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$KR" --output-format csv -d /tmp/pmc_r05a_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/pmc_fetch.log 2>&1
    timeout 1200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$KR" --output-format csv -d /tmp/pmc_r05a_write -o w -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/pmc_write.log 2>&1
    cd $R
    python scratch/r05_pmc_summary.py /tmp/pmc_r05a_sq $O/pmc_sq_in_situ.csv 30 > $O/pmc_sq_top.txt 2>&1; cut -c1-230 $O/pmc_sq_top.txt
    python scratch/r05_pmc_summary.py /tmp/pmc_r05a_fetch $O/pmc_fetch_in_situ.csv 30 > $O/pmc_fetch_top.txt 2>&1; head -12 $O/pmc_fetch_top.txt | cut -c1-200
    python scratch/r05_pmc_summary.py /tmp/pmc_r05a_write $O/pmc_write_in_situ.csv 30 > $O/pmc_write_top.txt 2>&1; head -12 $O/pmc_write_top.txt | cut -c1-200
    ;;
b)  # register-B GEMMs (csrc/gemm_rb.hip): bit-exactness against the round-4 kernels on awkward shapes, then the A/B on the step's dense shapes with COLD A operands
    L=$P/libfairdiff_hip_bench.so
    MB_RB_NOASSERT=1 FAIRDIFF_LIB=$L timeout 900 python scratch/mb_rb.py $2 2>&1 | grep -v amdgpu.ids > gpurun_out/r05b_mb_rb.txt; cat gpurun_out/r05b_mb_rb.txt
    ;;
c)  FAIRDIFF_LIB=$P/libfairdiff_hip_bench.so timeout 600 python scratch/dbg_rb.py 2>&1 | grep -v amdgpu.ids | cut -c1-220 ;;
*) echo "unknown pass $1";;
esac
