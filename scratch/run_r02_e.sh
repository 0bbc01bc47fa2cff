#!/bin/bash
# round-2 GPU pass E: fused QKV + strided attention + two-stream R1||R2: parity tests, then bench A/B
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
echo "== kernel tests (attention / transpose / fp8)"
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or transpose or elementwise" > gpurun_out/r02_e_kernels.log 2>&1; echo "rc=$?" >> gpurun_out/r02_e_kernels.log
tail -4 gpurun_out/r02_e_kernels.log | cut -c1-300
echo "== engine tests"
timeout 1800 python -m pytest tests/test_engine_gpu.py -q -x --durations=8 > gpurun_out/r02_e_engine.log 2>&1; echo "rc=$?" >> gpurun_out/r02_e_engine.log
grep -E "passed|failed|rc=|^E |Error|s call" gpurun_out/r02_e_engine.log | cut -c1-300 | tail -25
echo "== fullsize (unet + step)"
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -x -s -k "unet_cfg_pair or full_step" > gpurun_out/r02_e_full.log 2>&1; echo "rc=$?" >> gpurun_out/r02_e_full.log
grep -E "cosine|passed|failed|rc=|^E |SD15 step|SD15 unet eps" gpurun_out/r02_e_full.log | cut -c1-300 | tail -20
echo "== bench default"
timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > gpurun_out/r02_bench_e.json 2> gpurun_out/r02_bench_e.err; echo "rc=$?"; tail -2 gpurun_out/r02_bench_e.err | cut -c1-300
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02_bench_e.json")); print("default:", round(d["value"],3), "img/s", round(d["ms_per_step"],1), "ms", d["config"]["phase_ms"], d["roofline"]["kernel"], round(d["roofline"]["achieved"],1))
PY
echo "== bench FD_NO_CONCURRENT_R2"
FD_NO_CONCURRENT_R2=1 timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/r02_bench_e_noconc.json 2> gpurun_out/r02_bench_e_noconc.err; echo "rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02_bench_e_noconc.json")); print("no concurrent R2:", round(d["value"],3), "img/s", round(d["ms_per_step"],1), "ms")
PY
