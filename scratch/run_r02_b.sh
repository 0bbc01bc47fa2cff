#!/bin/bash
# round-2 GPU pass B: single-rank RCCL smoke, new parity tests, exp-4 OT overlap timing, MFMA peak calibration
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
echo "== rccl single-rank smoke (process group of 1 over RCCL)"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --force_collectives --experiment exp-4 --batch 4 --S 4 --tiny --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/r02_rccl_single.log 2>&1
echo "rc=$?" >> gpurun_out/r02_rccl_single.log
tail -3 gpurun_out/r02_rccl_single.log | cut -c1-1500
echo "== mfma peak"
timeout 300 ./scratch/mb_mfma_peak > gpurun_out/r02_mfma_peak.txt 2>&1; cat gpurun_out/r02_mfma_peak.txt
echo "== fullsize tests"
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -s --durations=10 > gpurun_out/r02_fullsize.log 2>&1
echo "rc=$?" >> gpurun_out/r02_fullsize.log
grep -E "^\[|cosine|oracle|passed|failed|Error|error|r=|rc=|^E " gpurun_out/r02_fullsize.log | cut -c1-300 | tail -150
echo "== new engine tests"
timeout 1500 python -m pytest tests/test_engine_gpu.py -q -s --durations=10 -k "multi_attribute_with_oracle or smooth_head or generate_image_matches or shared_mode or train_loop or train_driver" > gpurun_out/r02_engine_new.log 2>&1
echo "rc=$?" >> gpurun_out/r02_engine_new.log
grep -E "^\[|cosine|OT host|smooth|uint8|passed|failed|Error|error|rc=|^E " gpurun_out/r02_engine_new.log | cut -c1-300 | tail -80
echo "== exp-4 bench (OT overlap)"
timeout 600 python bench.py --experiment exp-4 --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/r02_bench_exp4.json 2> gpurun_out/r02_bench_exp4.err
echo "rc=$?"; tail -2 gpurun_out/r02_bench_exp4.err | cut -c1-300; cut -c1-2500 gpurun_out/r02_bench_exp4.json
