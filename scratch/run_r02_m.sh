#!/bin/bash
# round-2 GPU pass M: final profiles (PMC traffic by kernel, kernel trace), bench line, then the full CPU-baseline protocol
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
rm -rf gpurun_out/pmc_r02_fetch gpurun_out/pmc_r02_write
bash scratch/prof_pmc_r02.sh 2>&1 | tail -16
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02_c -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/gpurun_out/prof_r02_c.log 2>&1
cd $R
DB=$(find gpurun_out/prof_r02_c -name "*.db" | head -1)
python scratch/profsum.py $DB gpurun_out/r02_kernel_stats_c.csv 25
timeout 900 python bench.py --steps 4 --warmup 1 --dump_shapes gpurun_out/r02_gemm_shapes_final.csv > gpurun_out/r02_bench_m.json 2> gpurun_out/r02_bench_m.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_m.json')); print(round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['traffic'], d['roofline']['avg_launch_us'])"
timeout 2400 python bench.py --steps 2 --warmup 1 --no_roofline --cpu_baseline_full > gpurun_out/r02_bench_cpu_full.json 2> gpurun_out/r02_bench_cpu_full.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_cpu_full.json')); print(json.dumps(d['cpu_baseline'], indent=1)[:1500])"
