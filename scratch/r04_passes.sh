#!/bin/bash
# every gpurun pass of round 4 is one case branch:  bash scratch/r04_passes.sh <letter>
mkdir -p gpurun_out
P=$PWD/finetune_fair_diffusion_amd
B="python bench.py --no_cpu_baseline --no_roofline"
case "$1" in
a)  # the fixed library (no packed-fp32 VALU, multi-row LayerNorm backward shipped): race detectors, then whole-step A/B against the round-3 library
    timeout 300 python scratch/diag_hazard.py s3x3 delayx4 > gpurun_out/r04a_hazard_fixed_lib.txt 2>&1
    timeout 600 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "delay_injection or reproduce_themselves or shipped_schedule" > gpurun_out/r04a_race_tests.txt 2>&1
    for i in 1 2; do
      $B --steps 6 --warmup 2 > gpurun_out/r04a_bench_new_$i.json 2> gpurun_out/r04a_bench_new_$i.err
      FAIRDIFF_LIB=$P/libfairdiff_hip_slp_r03.so $B --steps 6 --warmup 2 > gpurun_out/r04a_bench_r03lib_$i.json 2> gpurun_out/r04a_bench_r03lib_$i.err
    done
    grep -h "differ" gpurun_out/r04a_hazard_fixed_lib.txt | sort | uniq -c | head; tail -5 gpurun_out/r04a_race_tests.txt
    python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04a_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],3), round(d['ms_per_step'],1), d['config']['phase_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
    ;;
b)  # full GPU suite + smoke on the library without packed-fp32 VALU (multi-row LayerNorm backward, gather warp backward)
    timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r04b_gpu_suite.log 2>&1; tail -25 gpurun_out/r04b_gpu_suite.log
    timeout 300 python __graft_entry__.py smoke > gpurun_out/r04b_smoke.log 2>&1; tail -2 gpurun_out/r04b_smoke.log
    ;;
c)  # which change moved the exp-3 tiny-model gradient cosine (0.981 in round 2 -> 0.966)?  same test under the A/B switches
    T="python -m pytest tests/test_engine_gpu.py -x -q -s -k test_full_step_multi_attribute_exp3"
    (echo "## default"; $T; echo "## FD_HOST_SCALES=1"; FD_HOST_SCALES=1 $T; echo "## round-3 library"; FAIRDIFF_LIB=$P/libfairdiff_hip_slp_r03.so $T; echo "## round-3 library + host scales"; FD_HOST_SCALES=1 FAIRDIFF_LIB=$P/libfairdiff_hip_slp_r03.so $T) 2>&1 | grep -i "##\|cosine\|passed\|failed" > gpurun_out/r04c_exp3_cosine_ab.txt
    cat gpurun_out/r04c_exp3_cosine_ab.txt
    ;;
d)  # device-resident tail (exp-1): suite subsets, then whole-step A/B host tail vs device tail, and the batch-16 / S=10 probe (same image-timesteps
    # per step as the headline: does a 2x larger backward batch pay?)
    timeout 1200 python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py tests/test_two_rank_gpu.py -q -s > gpurun_out/r04d_tests.log 2>&1; grep -i 'cosine\|passed\|failed\|FAILED' gpurun_out/r04d_tests.log | tail -60
    for i in 1 2; do
      $B --steps 6 --warmup 2 > gpurun_out/r04d_bench_devtail_$i.json 2> gpurun_out/r04d_bench_devtail_$i.err
      FD_HOST_TAIL=1 FD_HOST_SCALES=1 $B --steps 6 --warmup 2 > gpurun_out/r04d_bench_hosttail_$i.json 2> gpurun_out/r04d_bench_hosttail_$i.err
    done
    $B --steps 4 --warmup 2 --batch 16 --S 10 > gpurun_out/r04d_bench_b16_s10.json 2> gpurun_out/r04d_bench_b16_s10.err
    (FAIRDIFF_LIB=$P/libfairdiff_hip_lnmr.so timeout 200 python scratch/repro_packed_fp32_hazard.py 1500; timeout 200 python scratch/repro_packed_fp32_hazard.py 1500) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04d_repro_packed_fp32.txt; cat gpurun_out/r04d_repro_packed_fp32.txt
    python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04d_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],3), round(d['ms_per_step'],1), d['config']['phase_ms'], d['config']['host_ms_between_phase_marks'])
    except Exception as e: print(f, 'ERR', e)
PY
    ;;
e)  # 50-step soak at S = 50 (configs[3] rollout length), exp-4, all loss terms: allocator numbers per step
    timeout 900 python scratch/soak_s50.py 50 2>&1 | grep -v amdgpu.ids > gpurun_out/r04e_soak_s50_exp4.txt; tail -4 gpurun_out/r04e_soak_s50_exp4.txt
    ;;
f)  # schedule knobs re-measured with the device-resident tail: R2-side forwards on the R2 stream vs the launch stream, prefetch depth, backward streams
    for v in "" "FD_R2_SIDE_HOST_ORDER=1" "FD_R2_PREFETCH_STEPS=6" "FD_R2_PREFETCH_STEPS=10" "FD_R2_PREFETCH_STEPS=12" "FD_BWD_STREAMS=2" "FD_BWD_STREAMS=4" "" "FD_R2_SIDE_HOST_ORDER=1"; do
      n=$(echo "$v" | tr '=' '_'); [ -z "$n" ] && n=default
      env $v $B --steps 6 --warmup 2 > gpurun_out/r04f_${n}_$RANDOM.json 2>/dev/null
    done
    timeout 600 python -m pytest tests/test_two_rank_gpu.py -q -s -k eight > gpurun_out/r04f_eight_ranks.log 2>&1; grep -v "amdgpu.ids\|socket.cpp\|Gloo" gpurun_out/r04f_eight_ranks.log | tail -40
    python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04f_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],3), round(d['ms_per_step'],1), d['config']['phase_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
    ;;
g)  # deterministic shared dK / dV (slabs + per-timestep slots): tests, then whole-step A/B against the atomics form; late R2 prefetch
    timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_fullsize_gpu.py -q -x > gpurun_out/r04g_tests.log 2>&1; tail -6 gpurun_out/r04g_tests.log
    for v in "" "FD_ATOMIC_DKDV=1" "FD_R2_PREFETCH_LATE=2" "FD_R2_PREFETCH_LATE=4" "" "FD_ATOMIC_DKDV=1" "FD_R2_PREFETCH_LATE=3"; do
      n=$(echo "$v" | tr '=' '_'); [ -z "$n" ] && n=default
      env $v $B --steps 6 --warmup 2 > gpurun_out/r04g_${n}_$RANDOM.json 2>/dev/null
    done
    python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04g_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],3), round(d['ms_per_step'],1), d['config']['phase_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
    ;;
h)  # rocprofv3 kernel trace of the current tree (3-step bench) -> profiles/r04_bench_step_kernel_stats_*.csv
    O=gpurun_out/r04h; mkdir -p $O; R=$PWD
    cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_r04h -o r04h -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
    cd $R
    DB=$(find /tmp/prof_r04h -name "*.db" | head -1)
    python scratch/profsum.py $DB $O/kernel_stats.csv 45 > $O/kernel_stats_top.txt
    cut -c1-170 $O/kernel_stats_top.txt
    ;;
i)  # final validation: full GPU suite with durations, smoke, default bench line, 20-step bench line (runs the cfg1 CPU protocol itself)
    timeout 2400 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r04i_gpu_suite.log 2>&1; tail -22 gpurun_out/r04i_gpu_suite.log
    timeout 300 python __graft_entry__.py smoke > gpurun_out/r04i_smoke.log 2>&1; tail -1 gpurun_out/r04i_smoke.log
    timeout 900 python bench.py > gpurun_out/r04i_bench_default.json 2> gpurun_out/r04i_bench_default.err; cut -c1-400 gpurun_out/r04i_bench_default.json
    ;;
j)  timeout 2400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04j_bench_20_steps.json 2> gpurun_out/r04j_bench_20_steps.err; cut -c1-600 gpurun_out/r04j_bench_20_steps.json
    ;;
k)  # forward attention with 64 queries per wave (QB = 2): correctness under the kernel tests, isolated A/B, whole-step A/B (bench-hooks library, FD_ATTN_FWD_QB)
    L=$P/libfairdiff_hip_bench.so
    FAIRDIFF_LIB=$L FD_ATTN_FWD_QB=2 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" > gpurun_out/r04k_attn_tests_qb2.log 2>&1; tail -3 gpurun_out/r04k_attn_tests_qb2.log
    (echo "## QB=1"; FAIRDIFF_LIB=$L FD_ATTN_FWD_QB=1 python scratch/mb_attn_tr.py; echo "## QB=2"; FAIRDIFF_LIB=$L FD_ATTN_FWD_QB=2 python scratch/mb_attn_tr.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04k_attn_fwd_qb_isolated.txt; cat gpurun_out/r04k_attn_fwd_qb_isolated.txt
    for i in 1 2; do
      FAIRDIFF_LIB=$L FD_ATTN_FWD_QB=1 $B --steps 6 --warmup 2 > gpurun_out/r04k_step_qb1_$i.json 2>/dev/null
      FAIRDIFF_LIB=$L FD_ATTN_FWD_QB=2 $B --steps 6 --warmup 2 > gpurun_out/r04k_step_qb2_$i.json 2>/dev/null
    done
    python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04k_step_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],3), round(d['ms_per_step'],1), d['config']['phase_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
    ;;
l)  # attention forward: lazy softmax reference point (default) vs moving it with every maximum (-DFD_ATTN_LAZY=0), each at QB = 1 / 2; parity first
    L=$P/libfairdiff_hip_bench.so; N=$P/libfairdiff_hip_bench_nolazy.so
    for q in 1 2; do FAIRDIFF_LIB=$L FD_ATTN_FWD_QB=$q timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -1; done
    (for lib in $N $L; do for q in 1 2; do echo "## $(basename $lib) QB=$q"; FAIRDIFF_LIB=$lib FD_ATTN_FWD_QB=$q python scratch/mb_attn_tr.py 2>&1 | grep -v amdgpu.ids | head -4; done; done) > gpurun_out/r04l_attn_fwd_lazy_qb_isolated.txt; cat gpurun_out/r04l_attn_fwd_lazy_qb_isolated.txt
    for i in 1 2; do
      FAIRDIFF_LIB=$N FD_ATTN_FWD_QB=1 $B --steps 6 --warmup 2 > gpurun_out/r04l_step_nolazy_qb1_$i.json 2>/dev/null
      FAIRDIFF_LIB=$L FD_ATTN_FWD_QB=1 $B --steps 6 --warmup 2 > gpurun_out/r04l_step_lazy_qb1_$i.json 2>/dev/null
      FAIRDIFF_LIB=$L FD_ATTN_FWD_QB=2 $B --steps 6 --warmup 2 > gpurun_out/r04l_step_lazy_qb2_$i.json 2>/dev/null
    done
    python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04l_step_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],3), round(d['ms_per_step'],1), d['config']['phase_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
    ;;
m)  # the hazard, one level deeper: the SLP-vectorised norm.hip (packed fp32; everything else as shipped) under the pair detector -- as built,
    # with every s_waitcnt forced to zero, and with an s_nop 2 in front of every instruction; then 20 delay-injected runs of the shipped library
    for v in slp slp_wc0 slp_snop; do
      echo "## libfairdiff_hip_norm_$v.so"; FAIRDIFF_LIB=$P/libfairdiff_hip_norm_$v.so timeout 300 python scratch/diag_hazard3.py 8 2>&1 | grep -v amdgpu.ids | grep "pairs\|differing elements\|rows mod"
    done > gpurun_out/r04m_hazard_waitcnt_snop.txt; cat gpurun_out/r04m_hazard_waitcnt_snop.txt
    timeout 400 python scratch/diag_hazard.py delayx20 > gpurun_out/r04m_delay_x20_full.txt 2>&1
    (echo "delay-injected runs, per-stream buffers compared bitwise with the one-stream order: lines = 20 runs x 3 buffers"; echo "bit-identical buffers: $(grep '^\[delay' gpurun_out/r04m_delay_x20_full.txt | grep -c " 0 'rest' tensors differ (max rel 0.00e+00), 0 kv")"; echo "buffers with any difference: $(grep '^\[delay' gpurun_out/r04m_delay_x20_full.txt | grep -vc " 0 'rest' tensors differ (max rel 0.00e+00), 0 kv")") > gpurun_out/r04m_delay_x20_shipped.txt; cat gpurun_out/r04m_delay_x20_shipped.txt
    ;;
n)  # ViT-sized GEMMs of the tail (M = 2112 rows, N = 1280: 68 tiles of 128x320 -- below the 80-tile threshold, so they run on 64x64 tiles): lower the threshold
    L=$P/libfairdiff_hip_bench.so
    for i in 1 2; do
      for t in 80 64 48; do FAIRDIFF_LIB=$L FD_GEMM_T128=$t $B --steps 6 --warmup 2 > gpurun_out/r04n_t128_${t}_$i.json 2>/dev/null; done
    done
    python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04n_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],3), round(d['ms_per_step'],1), d['config']['phase_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
    ;;
o)  # hipGraph replay of the frozen model's forward: bit-identity, then whole-step A/B
    timeout 600 python scratch/check_r2_graph.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04o_r2_graph_check.txt; cat gpurun_out/r04o_r2_graph_check.txt | tail -12
    for i in 1 2; do
      $B --steps 6 --warmup 3 > gpurun_out/r04o_step_eager_$i.json 2>/dev/null
      FD_R2_GRAPH=1 $B --steps 6 --warmup 3 > gpurun_out/r04o_step_graph_$i.json 2> gpurun_out/r04o_step_graph_$i.err
    done
    python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04o_step_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],3), round(d['ms_per_step'],1), d['config']['phase_ms'], d['config']['host_ms_between_phase_marks'])
    except Exception as e: print(f, 'ERR', e)
PY
    tail -3 gpurun_out/r04o_step_graph_1.err
    ;;
p)  # GroupNorm statistics from the producers' epilogues (fd_gemm_desc.gn_stats): kernel tests, the U-Net / step tests, then whole-step A/B (FD_NO_GN_STATS=1)
    timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -s -k "groupnorm or statistics or gemm_big or conv3x3" > gpurun_out/r04p_kernel_tests.log 2>&1; grep -i "gn_stats\|passed\|failed\|Error" gpurun_out/r04p_kernel_tests.log | tail -30
    timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "unet or full_step or pair" > gpurun_out/r04p_engine_tests.log 2>&1; tail -5 gpurun_out/r04p_engine_tests.log
    for i in 1 2; do
      $B --steps 6 --warmup 3 > gpurun_out/r04p_step_stats_$i.json 2> gpurun_out/r04p_step_stats_$i.err
      FD_NO_GN_STATS=1 $B --steps 6 --warmup 3 > gpurun_out/r04p_step_nostats_$i.json 2>/dev/null
    done
    python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04p_step_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],3), round(d['ms_per_step'],1), d['config']['phase_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
    tail -3 gpurun_out/r04p_step_stats_1.err
    ;;
q)  # LayerNorm as a second output of the producing GEMM (fd_gemm_desc.ln_out): kernel tests, the U-Net / step tests, isolated cost, whole-step A/B (FD_NO_LN_EPILOGUE=1)
    timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -s -k "layernorm or gemm_big or statistics" > gpurun_out/r04q_kernel_tests.log 2>&1; grep -i "LayerNorm\|saved\|passed\|failed\|Error" gpurun_out/r04q_kernel_tests.log | tail -30
    timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "unet or full_step or pair" > gpurun_out/r04q_engine_tests.log 2>&1; tail -5 gpurun_out/r04q_engine_tests.log
    timeout 200 python scratch/mb_ln_epilogue.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04q_mb_ln_epilogue.txt
    for i in 1 2; do
      FD_LN_EPILOGUE=1 $B --steps 6 --warmup 3 > gpurun_out/r04q_step_ln_$i.json 2> gpurun_out/r04q_step_ln_$i.err
      $B --steps 6 --warmup 3 > gpurun_out/r04q_step_noln_$i.json 2>/dev/null
    done
    python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04q_step_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],3), round(d['ms_per_step'],1), d['config']['phase_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
    tail -3 gpurun_out/r04q_step_ln_1.err
    ;;
r)  # LayerNorm epilogue, second form (values back into the staging slot, segmented shuffle): whole-step A/B only
    B2="python bench.py --no_cpu_baseline --no_roofline"
    for i in 1 2 3; do
      FD_LN_EPILOGUE=1 $B2 --steps 6 --warmup 3 > gpurun_out/r04r_step_ln_$i.json 2> gpurun_out/r04r_step_ln_$i.err
      $B2 --steps 6 --warmup 3 > gpurun_out/r04r_step_noln_$i.json 2>/dev/null
    done
    python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04r_step_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],3), round(d['ms_per_step'],1), d['config']['phase_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
    ;;
s)  # pre-scaled q: kernel tests (attention, colscale), engine tests, isolated attention timing, whole-step A/B (FD_NO_PRESCALED_Q=1)
    timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or column_scale or gemm_big or gemm_plain" > gpurun_out/r04s_kernel_tests.log 2>&1; tail -4 gpurun_out/r04s_kernel_tests.log
    timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "unet or full_step or pair or generate" > gpurun_out/r04s_engine_tests.log 2>&1; tail -5 gpurun_out/r04s_engine_tests.log
    timeout 200 python scratch/mb_attn_pre.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04s_mb_attn_pre.txt
    for i in 1 2; do
      $B --steps 6 --warmup 3 > gpurun_out/r04s_step_pre_$i.json 2> gpurun_out/r04s_step_pre_$i.err
      FD_NO_PRESCALED_Q=1 $B --steps 6 --warmup 3 > gpurun_out/r04s_step_nopre_$i.json 2>/dev/null
    done
    python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04s_step_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],3), round(d['ms_per_step'],1), d['config']['phase_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
    tail -3 gpurun_out/r04s_step_pre_1.err
    ;;
t)  # GroupNorm statistics epilogue, cleaner whole-step A/B (three alternations) on the final tree
    for i in 1 2 3; do
      $B --steps 6 --warmup 3 > gpurun_out/r04t_step_stats_$i.json 2>/dev/null
      FD_NO_GN_STATS=1 $B --steps 6 --warmup 3 > gpurun_out/r04t_step_nostats_$i.json 2>/dev/null
    done
    python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04t_step_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],3), round(d['ms_per_step'],1), d['config']['phase_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
    ;;
u)  # step-time jitter: 20-step runs with and without the hipGraph replay of the frozen model's forward (does the freed host time remove the +50..95 ms outliers?)
    for i in 1 2; do
      $B --steps 20 --warmup 5 > gpurun_out/r04u_eager_$i.json 2>/dev/null
      FD_R2_GRAPH=1 $B --steps 20 --warmup 5 > gpurun_out/r04u_graph_$i.json 2> gpurun_out/r04u_graph_$i.err
    done
    python - <<'PY'
import json,glob,statistics
for f in sorted(glob.glob('gpurun_out/r04u_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); h=d['config']['host_ms_per_step']
        print(f, round(d['value'],3), round(d['ms_per_step'],1), 'median', round(statistics.median(h),1), 'outliers(>median+20):', sum(x>statistics.median(h)+20 for x in h), [round(x) for x in h])
    except Exception as e: print(f, 'ERR', e)
PY
    ;;
*) echo "unknown pass $1";;
esac
