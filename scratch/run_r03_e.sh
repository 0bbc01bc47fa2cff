#!/bin/bash
# round-3 GPU pass E: the complete -m gpu suite (no -x), PMC traffic passes for the dominant GEMM / conv kernels (incl. the ping-pong ones),
# the default bench line, the S = 50 exp-4 line (configs[3] rollout length), the bf16 / bf16 + e4m3 lines (configs[4] precision).
set -x
O=gpurun_out/r03e
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 2400 python -m pytest tests -m gpu -q --durations=12 ) > $O/pytest_all.log 2>&1
tail -25 $O/pytest_all.log
bash scratch/prof_pmc_r03.sh > $O/pmc.log 2>&1
tail -25 $O/pmc.log
cp profiles/r03_pmc_traffic.json $O/ 2>/dev/null
for d in pmc_r03_fetch pmc_r03_write; do f=$(find gpurun_out/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && head -c 3000000 $f > $O/$d.csv; done
rm -rf gpurun_out/pmc_r03_fetch gpurun_out/pmc_r03_write
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json
timeout 900 python bench.py --S 50 --experiment exp-4 --steps 3 --warmup 1 --no_cpu_baseline > $O/bench_exp4_s50.json 2> $O/bench_exp4_s50.err
tail -c 1500 $O/bench_exp4_s50.json
timeout 600 python bench.py --dtype bf16 --no_cpu_baseline --no_roofline > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 600 python bench.py --dtype bf16 --fp8_attn --no_cpu_baseline --no_roofline > $O/bench_bf16_fp8.json 2> $O/bench_bf16_fp8.err
for f in $O/bench_bf16.json $O/bench_bf16_fp8.json; do python -c "import sys,json; d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', round(d['value'],3), 'img/s', round(d['ms_per_step'],1))"; done
