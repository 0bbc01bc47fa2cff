#!/bin/bash
# round-end measurement set (run on the GPU box via gpurun): kernel trace + two PMC passes (HBM fetch / write) of the bench step
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof4 -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/gpurun_out/prof4.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 0 --no_cpu_baseline --no_roofline > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o w -- python3 $R/bench.py --steps 1 --warmup 0 --no_cpu_baseline --no_roofline > $R/gpurun_out/pmc_write.log 2>&1
ls -la $R/gpurun_out/prof4 $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write | head -40
tail -2 $R/gpurun_out/prof4.log | cut -c1-300
