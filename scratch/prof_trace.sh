#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof11 -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/gpurun_out/prof11.log 2>&1
tail -1 $R/gpurun_out/prof11.log | cut -c1-200
