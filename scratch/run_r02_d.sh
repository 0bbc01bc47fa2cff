#!/bin/bash
# round-2 GPU pass D: fp8 attention kernel band, bf16 library checks (+fp8 in-network), bf16 / bf16+fp8 bench lines, fixed tests of pass B
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
echo "== fp8 kernel tests"
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -s -k "fp8" > gpurun_out/r02_fp8_kernel.log 2>&1; echo "rc=$?" >> gpurun_out/r02_fp8_kernel.log
grep -E "^\[|fp8 attention|passed|failed|rc=|^E " gpurun_out/r02_fp8_kernel.log | cut -c1-250 | tail -40
echo "== bf16 checks"
timeout 1800 python -m pytest tests/test_bf16_gpu.py -q -s > gpurun_out/r02_bf16.log 2>&1; echo "rc=$?" >> gpurun_out/r02_bf16.log
grep -E "^\[bf16|cosine|targets|RMS|oracle|PASSED|passed|failed|rc=|^E |Error" gpurun_out/r02_bf16.log | cut -c1-250 | tail -80
echo "== re-run of the adjusted tests"
timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_engine_gpu.py -q -s -k "vae_decode_512 or smooth_head" > gpurun_out/r02_fix.log 2>&1; echo "rc=$?" >> gpurun_out/r02_fix.log
grep -E "^\[|smooth head|SD15 vae|passed|failed|rc=|^E " gpurun_out/r02_fix.log | cut -c1-300 | tail -30
echo "== bench bf16"
timeout 600 python bench.py --dtype bf16 --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/r02_bench_bf16.json 2> gpurun_out/r02_bench_bf16.err; echo "rc=$?"; tail -2 gpurun_out/r02_bench_bf16.err | cut -c1-300; cut -c1-1200 gpurun_out/r02_bench_bf16.json
echo "== bench bf16 + fp8 attention"
timeout 600 python bench.py --dtype bf16 --fp8_attn --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/r02_bench_bf16_fp8.json 2> gpurun_out/r02_bench_bf16_fp8.err; echo "rc=$?"; tail -2 gpurun_out/r02_bench_bf16_fp8.err | cut -c1-300; cut -c1-1200 gpurun_out/r02_bench_bf16_fp8.json
