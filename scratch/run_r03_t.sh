#!/bin/bash
# round-3 GPU pass T: final validation checkpoint (8 hardware queues, transpose-read attention, device OT solver): full -m gpu suite,
# default bench invocation, rocprofv3 kernel stats of a 3-step bench.
set -x
O=gpurun_out/r03t
mkdir -p $O
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.err
python -c "import sys,json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['roofline']['frac'], d['cpu_baseline'])"
R=$PWD
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o r03t -- python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
cd $R
DB=$(find /tmp/prof_t -name "*.db" | head -1)
python scratch/profsum.py $DB $O/kernel_stats.csv 40 > $O/kernel_stats_top.txt
cat $O/kernel_stats_top.txt | cut -c1-160
