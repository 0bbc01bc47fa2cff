#!/bin/bash
# round-2 GPU pass A: RCCL smoke on a shared GPU, full-size parity tests, default bench
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
echo "== rccl smoke" 
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --share_gpu0 --batch 2 --S 4 --tiny --steps 2 --warmup 1 --no_cpu_baseline > gpurun_out/r02_rccl_smoke.log 2>&1
echo "rc=$?" >> gpurun_out/r02_rccl_smoke.log
tail -5 gpurun_out/r02_rccl_smoke.log | cut -c1-600
echo "== fullsize tests"
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -x -q -s --durations=10 > gpurun_out/r02_fullsize.log 2>&1
echo "rc=$?" >> gpurun_out/r02_fullsize.log
grep -E "^\[|cosine|oracle|passed|failed|Error|error|r=|rc=" gpurun_out/r02_fullsize.log | cut -c1-300 | tail -120
echo "== bench"
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err
echo "rc=$?"; tail -3 gpurun_out/r02_bench_a.err | cut -c1-400
cat gpurun_out/r02_bench_a.json | cut -c1-6000
