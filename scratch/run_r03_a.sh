#!/bin/bash
# round-3 GPU pass A: goldens that need >62 GB of host RAM (oracle autograd at SD-v1.5 size) are generated on the GPU box's host CPU,
# then the whole -m gpu suite, the default bench line, and a rocprofv3 kernel trace of two bench steps.
set -x
O=gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
nproc; free -g | head -2
(python tests/golden/make_oracle_step_golden.py smooth cfg0 > $O/golden.log 2>&1; cp tests/golden/oracle_sd15_*.npz $O/) &
GP=$!
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest_tiny.log
wait $GP
cat $O/golden.log
ls -la tests/golden/
timeout 2400 python -m pytest tests/test_fullsize_gpu.py tests/test_two_rank_gpu.py tests/test_bf16_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -250 > $O/pytest_full.log
tail -30 $O/pytest_full.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o r03a -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err
cd $GRAFT_REPO_ROOT
find /tmp/prof_a -name "*kernel_stats*" | head
cp $(find /tmp/prof_a -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
head -40 $O/kernel_stats.csv
