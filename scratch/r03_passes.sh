#!/bin/bash
# The gpurun passes of round 3, one function per pass (a .. u), in the order they ran.  Usage on the GPU box: bash scratch/r03_passes.sh <letter>
# Each pass writes under gpurun_out/r03<letter>/; the summaries that are cited were copied to profiles/ by hand.
set -x
export TMPDIR=/tmp
case "$1" in
a)
  # round-3 GPU pass A: goldens that need >62 GB of host RAM (oracle autograd at SD-v1.5 size) are generated on the GPU box's host CPU,
  # then the whole -m gpu suite, the default bench line, and a rocprofv3 kernel trace of two bench steps.
  O=gpurun_out/r03a
  mkdir -p $O
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
  ;;
b)
  # round-3 GPU pass B: the two schedule-property tests after the tolerance split, the ping-pong GEMM A/B (bench-hooks library), GroupNorm
  # microbenchmark, CPU-oracle thread scaling, and a rocprofv3 kernel trace of two bench steps summarised with scratch/profsum.py.
  O=gpurun_out/r03b
  mkdir -p $O
  R=$GRAFT_REPO_ROOT
  timeout 1200 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -k "shipped_schedule or mixed_keep or golden" 2>&1 | grep -v "^$" | tail -80 > $O/pytest_sched.log
  grep -n "rel max err\|cosine\|passed\|failed\|kept" $O/pytest_sched.log | tail -30
  FAIRDIFF_LIB=$R/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so timeout 900 python scratch/mb_pp.py > $O/mb_pp.txt 2>&1
  cat $O/mb_pp.txt
  timeout 600 python scratch/mb_gn.py > $O/mb_gn.txt 2>&1
  tail -30 $O/mb_gn.txt
  timeout 900 python scratch/mb_cpu_threads.py > $O/cpu_threads.txt 2>&1
  cat $O/cpu_threads.txt
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o r03b -- python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
  cd $R
  DB=$(find /tmp/prof_b -name "*.db" | head -1)
  python scratch/profsum.py $DB $O/kernel_stats.csv 45 > $O/kernel_stats_top.txt
  cat $O/kernel_stats_top.txt
  ;;
c)
  # round-3 GPU pass C: ping-pong kernels incl. the 128x320 / split-K variants (microbench A/B), whole-step A/B of the dispatch policies through the
  # bench-hooks library, the kernel + full-size suites with the shipped policy, the default bench line and a kernel trace.
  O=gpurun_out/r03c
  mkdir -p $O
  R=$GRAFT_REPO_ROOT
  BL=$R/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
  FAIRDIFF_LIB=$BL timeout 900 python scratch/mb_pp.py > $O/mb_pp.txt 2>&1
  cat $O/mb_pp.txt
  for mode in 0 37 45 61 63; do
    FAIRDIFF_LIB=$BL FD_GEMM_PP=$mode timeout 600 python bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('PP_MODE $mode', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])" | tee -a $O/step_ab.txt
  done
  timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest_kernels.log
  cat $O/pytest_kernels.log
  timeout 2400 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -150 > $O/pytest_full.log
  grep -n "rel max err\|cosine\|passed\|failed\|kept\|Error" $O/pytest_full.log | cut -c1-230 | tail -60
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
  tail -c 1500 $O/bench.json
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o r03c -- python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
  cd $R
  DB=$(find /tmp/prof_c -name "*.db" | head -1)
  python scratch/profsum.py $DB $O/kernel_stats.csv 40 > $O/kernel_stats_top.txt
  cat $O/kernel_stats_top.txt
  ;;
d)
  # round-3 GPU pass D: the complete -m gpu suite with durations (budget: < 900 s), backward-stream and tile-threshold A/Bs of the whole step
  # now that the ping-pong kernels carry the convolutions, the default bench line, a kernel trace.
  O=gpurun_out/r03d
  mkdir -p $O
  R=$GRAFT_REPO_ROOT
  BL=$R/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
  ( time timeout 2400 python -m pytest tests -m gpu -q -x --durations=30 ) > $O/pytest_all.log 2>&1
  tail -50 $O/pytest_all.log
  one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"; }
  for st in 2 3 4; do
    FD_BWD_STREAMS=$st timeout 600 python bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "BWD_STREAMS=$st" | tee -a $O/step_ab.txt
  done
  for th in "100 80" "64 48" "140 110" "200 160"; do
    set -- $th
    FAIRDIFF_LIB=$BL FD_GEMM_T256=$1 FD_GEMM_T128=$2 timeout 600 python bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "T256=$1,T128=$2" | tee -a $O/step_ab.txt
  done
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
  tail -c 1200 $O/bench.json
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o r03d -- python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
  cd $R
  DB=$(find /tmp/prof_d -name "*.db" | head -1)
  python scratch/profsum.py $DB $O/kernel_stats.csv 40 > $O/kernel_stats_top.txt
  cat $O/kernel_stats_top.txt
  timeout 600 python scratch/mb_cpu_threads.py > $O/cpu_threads.txt 2>&1
  cat $O/cpu_threads.txt
  ;;
e)
  # round-3 GPU pass E: the complete -m gpu suite (no -x), PMC traffic passes for the dominant GEMM / conv kernels (incl. the ping-pong ones),
  # the default bench line, the S = 50 exp-4 line (configs[3] rollout length), the bf16 / bf16 + e4m3 lines (configs[4] precision).
  O=gpurun_out/r03e
  mkdir -p $O
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
  ;;
f)
  # round-3 GPU pass F: R2 prefetch under the tail -- equivalence test, whole-step A/B over the number of prefetched denoising steps.
  O=gpurun_out/r03f
  mkdir -p $O
  timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "prefetch or full_fairness_step or exp2 or train_loop" 2>&1 | tail -15 > $O/pytest.log
  cat $O/pytest.log
  one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['r2_steps_prefetched_under_previous_tail'], d['config']['phase_ms'], d['config']['host_ms_per_step'])"; }
  for k in 0 4 6 8 10 0 6; do
    FD_R2_PREFETCH_STEPS=$k timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>$O/err_$k.txt | one "R2_PREFETCH=$k" | tee -a $O/step_ab.txt
  done
  ;;
g)
  # round-3 GPU pass G: step-time jitter diagnosis (with / without the Python GC), tests touched by the LoRA-refresh and prefetch changes, bench.
  O=gpurun_out/r03g
  mkdir -p $O
  timeout 600 python scratch/diag_step_jitter.py > $O/jitter.txt 2>&1
  cat $O/jitter.txt | cut -c1-330
  DIAG_NOGC=1 timeout 600 python scratch/diag_step_jitter.py > $O/jitter_nogc.txt 2>&1
  grep "^step" $O/jitter_nogc.txt
  timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -6 > $O/pytest.log
  cat $O/pytest.log
  timeout 900 python bench.py --steps 8 --warmup 2 --no_cpu_baseline > $O/bench.json 2> $O/bench.err
  python -c "import sys,json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'], d['config']['host_ms_per_step'])"
  ;;
h)
  # round-3 GPU pass H: lagged gradient scales (no mid-step host syncs) -- tests and same-box A/B against FD_SYNC_SCALES=1.
  O=gpurun_out/r03h
  mkdir -p $O
  timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -q -x 2>&1 | tail -6 > $O/pytest_engine.log
  cat $O/pytest_engine.log
  timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_two_rank_gpu.py -m gpu -q -x -k "shipped or mixed or golden or two_rank" 2>&1 | tail -6 > $O/pytest_full.log
  cat $O/pytest_full.log
  one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'], d['config']['host_ms_per_step'])"; }
  for m in sync lagged sync lagged; do
    if [ $m = sync ]; then export FD_SYNC_SCALES=1; else unset FD_SYNC_SCALES; fi
    timeout 600 python bench.py --steps 8 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "SCALES=$m" | tee -a $O/step_ab.txt
  done
  ;;
i)
  # round-3 GPU pass I: validation checkpoint after the R2 prefetch / batched LoRA refresh / reverted lagged scales: full -m gpu suite,
  # default bench invocation, rocprofv3 kernel stats of a 3-step bench.
  O=gpurun_out/r03i
  mkdir -p $O
  timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/pytest_gpu.log
  cat $O/pytest_gpu.log
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
  tail -c 600 $O/bench.err
  python -c "import sys,json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['roofline']['frac'], d['cpu_baseline'])"
  R=$PWD
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_i -o r03i -- python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
  cd $R
  DB=$(find /tmp/prof_i -name "*.db" | head -1)
  python scratch/profsum.py $DB $O/kernel_stats.csv 40 > $O/kernel_stats_top.txt
  cat $O/kernel_stats_top.txt | cut -c1-160
  ;;
j)
  # round-3 GPU pass J: the new device OT solver test + persistent streaming GEMM: correctness (bit-equal to the shipped kernels) and isolated A/B.
  O=gpurun_out/r03j
  mkdir -p $O
  timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "ot_assign" 2>&1 | tail -5 > $O/pytest_ot.log
  cat $O/pytest_ot.log
  export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
  FD_GEMM_PPS_MIN=1 timeout 600 python scratch/mb_pps.py > $O/mb_pps_min1.txt 2>&1
  cat $O/mb_pps_min1.txt | cut -c1-200
  FD_GEMM_PPS_MIN=1 FD_GEMM_PPS_WG=512 timeout 600 python scratch/mb_pps.py > $O/mb_pps_wg512.txt 2>&1
  tail -32 $O/mb_pps_wg512.txt | cut -c1-200
  ;;
k)
  # round-3 GPU pass K: whole-step A/B of the persistent streaming GEMM policy (bench-hooks library, same box).
  O=gpurun_out/r03k
  mkdir -p $O
  export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
  one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"; }
  run() { FD_GEMM_PP=$1 FD_GEMM_PPS_MIN=$2 FD_GEMM_PPS_MAXK=$3 timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "PP=$1 PPS_MIN=$2 MAXK=$3" | tee -a $O/step_ab.txt; }
  run 45 512 100000
  run 173 1 100000
  run 173 256 100000
  run 173 512 100000
  run 45 512 100000
  run 173 1 700
  run 173 257 100000
  ;;
l)
  # round-3 GPU pass L: attention kernels with LDS transpose reads: parity (both forms, bit-identity), isolated A/B, whole-step A/B.
  O=gpurun_out/r03l
  mkdir -p $O
  timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" 2>&1 | tail -8 > $O/pytest_attn.log
  cat $O/pytest_attn.log
  timeout 600 python scratch/mb_attn_tr.py > $O/mb_attn_tr.txt 2>&1
  cat $O/mb_attn_tr.txt | cut -c1-250
  one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"; }
  for m in tr copies tr copies; do
    if [ $m = copies ]; then export FD_ATTN_NO_TR=1; else unset FD_ATTN_NO_TR; fi
    timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "ATTN=$m" | tee -a $O/step_ab.txt
  done
  ;;
m)
  # round-3 GPU pass M: dK/dV transpose-read form at three waves per SIMD (bench-hooks library built with -DFD_DKDV_TR_W3) vs two; step A/B.
  O=gpurun_out/r03m
  mkdir -p $O
  timeout 600 python scratch/mb_attn_tr.py > $O/mb_attn_w2.txt 2>&1
  grep "^B" $O/mb_attn_w2.txt | cut -c1-250
  export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
  timeout 600 python scratch/mb_attn_tr.py > $O/mb_attn_w3.txt 2>&1
  grep "^B" $O/mb_attn_w3.txt | cut -c1-250
  one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"; }
  for m in w3 w2 w3 w2; do
    if [ $m = w3 ]; then export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so; else unset FAIRDIFF_LIB; fi
    timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "DKDV=$m" | tee -a $O/step_ab.txt
  done
  ;;
n)
  # round-3 GPU pass N: dK/dV kernel with the second score tile's softmax interleaved with the first tile's dV / dK products (bench-hooks
  # library built with -DFD_DKDV_PIPE) vs the shipped order; parity on the variant; step A/B.
  O=gpurun_out/r03n
  mkdir -p $O
  timeout 600 python scratch/mb_attn_tr.py > $O/mb_attn_base.txt 2>&1
  grep "^B" $O/mb_attn_base.txt | cut -c1-250
  export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
  timeout 600 python scratch/mb_attn_tr.py > $O/mb_attn_pipe.txt 2>&1
  grep "^B" $O/mb_attn_pipe.txt | cut -c1-250
  timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention_fwd_bwd" 2>&1 | tail -3 > $O/pytest_attn_pipe.log
  cat $O/pytest_attn_pipe.log
  one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"; }
  for m in pipe base pipe base; do
    if [ $m = pipe ]; then export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so; else unset FAIRDIFF_LIB; fi
    timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "DKDV=$m" | tee -a $O/step_ab.txt
  done
  ;;
o)
  # round-3 GPU pass O: dQ kernel with the key mask behind a wave-uniform branch (product library) and attention built without SLP
  # vectorisation (bench-hooks library, -fno-slp-vectorize: the guide prices v_pk_*_f32 beside MFMAs as an anti-lever); parity; step A/B.
  O=gpurun_out/r03o
  mkdir -p $O
  timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" 2>&1 | tail -3 > $O/pytest_attn.log
  cat $O/pytest_attn.log
  timeout 600 python scratch/mb_attn_tr.py > $O/mb_attn_base.txt 2>&1
  grep "^B" $O/mb_attn_base.txt | cut -c1-250
  export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
  timeout 600 python scratch/mb_attn_tr.py > $O/mb_attn_noslp.txt 2>&1
  grep "^B" $O/mb_attn_noslp.txt | cut -c1-250
  one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"; }
  for m in noslp base noslp base; do
    if [ $m = noslp ]; then export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so; else unset FAIRDIFF_LIB; fi
    timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "ATTN=$m" | tee -a $O/step_ab.txt
  done
  ;;
p)
  # round-3 GPU pass P: validation checkpoint after the transpose-read attention, device OT solver, dQ mask branch: full -m gpu suite,
  # default bench invocation, rocprofv3 kernel stats of a 3-step bench.
  O=gpurun_out/r03p
  mkdir -p $O
  timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/pytest_gpu.log
  cat $O/pytest_gpu.log
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
  tail -c 600 $O/bench.err
  python -c "import sys,json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['roofline']['frac'], d['cpu_baseline'])"
  R=$PWD
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o r03p -- python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
  cd $R
  DB=$(find /tmp/prof_p -name "*.db" | head -1)
  python scratch/profsum.py $DB $O/kernel_stats.csv 40 > $O/kernel_stats_top.txt
  cat $O/kernel_stats_top.txt | cut -c1-160
  ;;
q)
  # round-3 GPU pass Q: configs[4] lines after the transpose-read attention (bf16 vs bf16 + e4m3 self-attention), exp-4 step with the device
  # OT solver vs the host solver, RCCL collectives at world size 1.
  O=gpurun_out/r03q
  mkdir -p $O
  timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
  timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --dtype bf16 --fp8_attn > $O/bench_bf16_fp8.json 2> $O/bench_bf16_fp8.err
  timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --dtype bf16 > $O/bench_bf16_b.json 2> $O/bench_bf16_b.err
  timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --dtype bf16 --fp8_attn > $O/bench_bf16_fp8_b.json 2> $O/bench_bf16_fp8_b.err
  timeout 600 python bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline --experiment exp-4 > $O/bench_exp4_device_ot.json 2> $O/bench_exp4_device_ot.err
  FD_OT_HOST=1 timeout 600 python bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline --experiment exp-4 > $O/bench_exp4_host_ot.json 2> $O/bench_exp4_host_ot.err
  timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --force_collectives > $O/bench_rccl_ws1.json 2> $O/bench_rccl_ws1.err
  for f in $O/*.json; do python -c "import sys,json; d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); c=d['config']; print('$f', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', c.get('ot_solver'), c.get('ot_solve_ms'), c.get('ot_exposed_wait_ms'))"; done
  ;;
r)
  # round-3 GPU pass R: number of HIP hardware queues (GPU_MAX_HW_QUEUES, default 4) vs the step's streams (launch, R2, two more backward
  # streams, OT, RCCL's internal one): streams that share a hardware queue serialise.  Whole-step A/B, with and without collectives.
  O=gpurun_out/r03r
  mkdir -p $O
  one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"; }
  for q in 4 8 4 8 2 16; do
    GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "HWQ=$q" | tee -a $O/step_ab.txt
  done
  for q in 4 8; do
    GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline --force_collectives 2>/dev/null | one "HWQ=$q collectives" | tee -a $O/step_ab.txt
  done
  timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "single_image_step or multi_attribute_with_oracle_ot" 2>&1 | tail -4 > $O/pytest.log
  cat $O/pytest.log
  ;;
s)
  # round-3 GPU pass S: with 8 hardware queues (now the package default), re-tune the stream knobs that were tuned under 4: backward streams, R2 prefetch depth.
  O=gpurun_out/r03s
  mkdir -p $O
  one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['hip_hw_queues'], d['config']['phase_ms'])"; }
  run() { env $1 timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "$1" | tee -a $O/step_ab.txt; }
  run FD_X=0
  run FD_BWD_STREAMS=4
  run FD_R2_PREFETCH_STEPS=12
  run FD_R2_PREFETCH_STEPS=16
  run FD_X=0
  run FD_BWD_STREAMS=2
  run FD_R2_PREFETCH_STEPS=20
  run GPU_MAX_HW_QUEUES=12
  ;;
t)
  # round-3 GPU pass T: final validation checkpoint (8 hardware queues, transpose-read attention, device OT solver): full -m gpu suite,
  # default bench invocation, rocprofv3 kernel stats of a 3-step bench.
  O=gpurun_out/r03t
  mkdir -p $O
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
  ;;
u)
  # round-3 GPU pass U: small host -> device copies of the loss phase through pinned memory (non-blocking) vs pageable (FD_NO_PINNED_H2D=1); engine tests.
  O=gpurun_out/r03u
  mkdir -p $O
  one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'], d['config']['host_ms_between_phase_marks'])"; }
  for m in pinned pageable pinned pageable; do
    if [ $m = pageable ]; then export FD_NO_PINNED_H2D=1; else unset FD_NO_PINNED_H2D; fi
    timeout 600 python bench.py --steps 8 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "H2D=$m" | tee -a $O/step_ab.txt
  done
  unset FD_NO_PINNED_H2D
  timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest_engine.log
  cat $O/pytest_engine.log
  ;;
v)
  # round-3 GPU pass V: 20-step bench (the driver's usual step count) as a soak of the final stream setup, then the 8(d) CPU-baseline protocol.
  O=gpurun_out/r03v
  mkdir -p $O
  timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_20.json 2> $O/bench_20.err
  python -c "import sys,json; d=json.loads([l for l in open('$O/bench_20.json') if l.startswith('{')][-1]); print(round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['host_ms_per_step'], d['config']['peak_hbm_gb'])"
  timeout 2400 python bench.py --steps 20 --warmup 2 --cpu_baseline_full --no_roofline > $O/bench_cpu_full.json 2> $O/bench_cpu_full.err
  python -c "import sys,json; d=json.loads([l for l in open('$O/bench_cpu_full.json') if l.startswith('{')][-1]); print(round(d['value'],3), 'img/s', d['cpu_baseline'])"
  ;;
w)
  # round-3 GPU pass W: the tail's enqueues reordered (classifier + recorded CLIP / DINO forward of R1's images before the first read-back; R2's feature
  # encoders before its logits' read-back) vs the old order (FD_NO_TAIL_REORDER=1); engine + full-size step tests; one run with fine phase marks.
  O=gpurun_out/r03w
  mkdir -p $O
  timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest_engine.log
  cat $O/pytest_engine.log
  one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'], d['config']['host_ms_between_phase_marks'])"; }
  for m in reorder old reorder old reorder old; do
    if [ $m = old ]; then export FD_NO_TAIL_REORDER=1; else unset FD_NO_TAIL_REORDER; fi
    timeout 600 python bench.py --steps 8 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "TAIL=$m" | tee -a $O/step_ab.txt
  done
  unset FD_NO_TAIL_REORDER
  FD_FINE_MARKS=1 timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "FINE" | tee -a $O/fine_marks.txt
  timeout 1700 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k "full_step or golden" 2>&1 | tail -4 > $O/pytest_fullsize.log
  cat $O/pytest_fullsize.log
  ;;
x)
  # round-3 GPU pass X: fine phase marks inside the face-realism branch of the loss (shipped tail order), twice.
  O=gpurun_out/r03x
  mkdir -p $O
  one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'], d['config']['host_ms_between_phase_marks'])"; }
  FD_FINE_MARKS=1 timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "FINE" | tee -a $O/fine_marks.txt
  FD_FINE_MARKS=1 timeout 600 python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_roofline 2>/dev/null | one "FINE" | tee -a $O/fine_marks.txt
  ;;
y)
  # round-3 GPU pass Y: host profile of the recorded CLIP / DINO forward (scratch/diag_recorded_features.py).
  O=gpurun_out/r03y
  mkdir -p $O
  timeout 900 python scratch/diag_recorded_features.py > $O/diag.txt 2>&1
  tail -80 $O/diag.txt | cut -c1-200
  ;;
z)
  # round-3 GPU pass Z: HIP stream priority of the frozen-model rollout (R2 + prefetch) stream: does a low-priority prefetch stop slowing the tail's small kernels?
  O=gpurun_out/r03z
  mkdir -p $O
  python -c "import torch; print('priority range (least, greatest):', torch.cuda.Stream.priority_range())" 2>&1 | tail -1 | tee $O/priority_range.txt
  one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"; }
  for pr in 0 1 -1 0 1 -1; do
    FD_R2_PRIORITY=$pr timeout 600 python bench.py --steps 8 --warmup 2 --no_cpu_baseline --no_roofline 2>$O/err_$pr.txt | one "R2_PRIORITY=$pr" | tee -a $O/step_ab.txt
  done
  tail -3 $O/err_1.txt
  ;;
A)
  # round-3 GPU pass AA: the whole step launched from a high-priority stream (FD_MAIN_PRIORITY=-1) vs the default stream; engine tests under it.
  O=gpurun_out/r03aa
  mkdir -p $O
  one() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['config']['phase_ms'])"; }
  for pr in 0 -1 0 -1 0 -1; do
    FD_MAIN_PRIORITY=$pr timeout 600 python bench.py --steps 8 --warmup 2 --no_cpu_baseline --no_roofline 2>$O/err.txt | one "MAIN_PRIORITY=$pr" | tee -a $O/step_ab.txt
  done
  tail -3 $O/err.txt
  FD_MAIN_PRIORITY=-1 timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest_engine.log
  cat $O/pytest_engine.log
  ;;
B)
  # round-3 GPU pass BB: final full -m gpu suite and the driver's smoke() on the final tree.
  O=gpurun_out/r03bb
  mkdir -p $O
  timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/pytest_gpu.log
  cat $O/pytest_gpu.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
  tail -3 $O/smoke.log
  ;;
C)
  # round-3 GPU pass CC: row padding of the backward attention kernels' row-major tiles (read as 16-byte fragments AND through transpose reads):
  # shipped 8 halfs vs 16 / 24 / 40 (bench-hooks libraries built with -DFD_ATTN_BWD_PAD=n).
  O=gpurun_out/r03cc
  mkdir -p $O
  for pad in 8 16 24 40 8; do
    if [ $pad = 8 ]; then unset FAIRDIFF_LIB; else export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench_pad$pad.so; fi
    echo "# pad $pad" | tee -a $O/mb_attn_pad.txt
    timeout 600 python scratch/mb_attn_tr.py 2>&1 | grep "^B" | cut -c1-250 | tee -a $O/mb_attn_pad.txt
  done
  ;;
D)
  # round-3 GPU pass DD: forward attention with the softmax denominator taken from a ones column of V (row D of the P.V accumulator) vs summed on the
  # VALU (bench-hooks library built with -DFD_ATTN_NO_ONES); parity of both forms; isolated forward A/B; engine tests.
  O=gpurun_out/r03dd
  mkdir -p $O
  timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" 2>&1 | tail -3 > $O/pytest_attn.log
  cat $O/pytest_attn.log
  for m in ones valu ones valu; do
    if [ $m = ones ]; then unset FAIRDIFF_LIB; else export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so; fi
    echo "# $m" | tee -a $O/mb_attn_ones.txt
    timeout 600 python scratch/mb_attn_tr.py 2>&1 | grep "^B" | cut -c1-110 | tee -a $O/mb_attn_ones.txt
  done
  unset FAIRDIFF_LIB
  timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_bf16_gpu.py -m gpu -q -x 2>&1 | tail -3 > $O/pytest_engine.log
  cat $O/pytest_engine.log
  ;;
E)
  # round-3 GPU pass EE: last full -m gpu suite, smoke() and default bench line on the final tree (after the forward-attention denominator change).
  O=gpurun_out/r03ee
  mkdir -p $O
  timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/pytest_gpu.log
  cat $O/pytest_gpu.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
  tail -2 $O/smoke.log
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
  python -c "import sys,json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['roofline']['frac'])"
  ;;
F)
  # round-3 GPU pass FF: rocprofv3 kernel trace of the final tree (3-step bench) -> profiles/r03_bench_step_kernel_stats_v7_final.csv
  O=gpurun_out/r03ff
  mkdir -p $O
  R=$PWD
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_ff -o r03ff -- python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
  cd $R
  DB=$(find /tmp/prof_ff -name "*.db" | head -1)
  python scratch/profsum.py $DB $O/kernel_stats.csv 40 > $O/kernel_stats_top.txt
  cat $O/kernel_stats_top.txt | cut -c1-160
  ;;
G)
  # round-3 GPU pass GG: LayerNorm with R rows per wave and gamma / beta in registers (product library) vs one row per wave (bench-hooks library built
  # with -DFD_LN_ONE_ROW): timing and output checksums (bit-identity), kernel tests.
  O=gpurun_out/r03gg
  mkdir -p $O
  echo "# shipped (R rows per wave)" | tee $O/mb_ln.txt
  timeout 300 python scratch/mb_ln.py 2>&1 | grep "^LN" | tee -a $O/mb_ln.txt
  echo "# one row per wave" | tee -a $O/mb_ln.txt
  FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so timeout 300 python scratch/mb_ln.py 2>&1 | grep "^LN" | tee -a $O/mb_ln.txt
  timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -3 > $O/pytest_kernels.log
  cat $O/pytest_kernels.log
  ;;
H)
  # round-3 GPU pass HH: full -m gpu suite, smoke and two bench runs after the LayerNorm change.
  O=gpurun_out/r03hh
  mkdir -p $O
  timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/pytest_gpu.log
  cat $O/pytest_gpu.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
  tail -2 $O/smoke.log
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
  python -c "import sys,json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(round(d['value'],3), 'img/s', round(d['ms_per_step'],1), 'ms', d['roofline']['frac'], d['config']['phase_ms'])"
  ;;
I)
  # round-3 GPU pass II: the B = 8 / S = 20 schedule test alone, with its output (it failed once in pass HH).
  O=gpurun_out/r03ii
  mkdir -p $O
  timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -s -k "shipped_schedule_equals_reference" 2>&1 | grep -v amdgpu.ids | tail -40 > $O/pytest_sched.log
  cat $O/pytest_sched.log | cut -c1-300
  ;;
J)
  # round-3 GPU pass JJ: row-position dependence of the LayerNorm kernel (scratch/diag_ln_rowpos.py).
  O=gpurun_out/r03jj
  mkdir -p $O
  timeout 300 python scratch/diag_ln_rowpos.py 2>&1 | grep -v amdgpu.ids | tee $O/diag.txt
  ;;
K)
  # round-3 GPU pass KK: the schedule test with the one-row-per-wave LayerNorm (bench-hooks library built with -DFD_LN_ONE_ROW) and with the shipped one, twice each.
  O=gpurun_out/r03kk
  mkdir -p $O
  for m in onerow shipped onerow shipped; do
    if [ $m = onerow ]; then export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so; else unset FAIRDIFF_LIB; fi
    echo "# $m" | tee -a $O/sched.txt
    timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -s -k "shipped_schedule_equals_reference" 2>&1 | grep -E "rel max err|passed|failed" | cut -c1-200 | tee -a $O/sched.txt
  done
  ;;
L)
  # round-3 GPU pass LL: which half of the multi-row LayerNorm breaks the schedule test: forward one-row / backward multi-row, and the reverse.
  O=gpurun_out/r03ll
  mkdir -p $O
  for m in fwd1 bwd1; do
    export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench_$m.so
    echo "# $m" | tee -a $O/sched.txt
    timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -s -k "shipped_schedule_equals_reference" 2>&1 | grep -E "rel max err|passed|failed" | cut -c1-200 | tee -a $O/sched.txt
  done
  ;;
M)
  # round-3 GPU pass MM: after moving the CFG-pair upstream gradient in front of the side streams' wait: the schedule test (shipped multi-row LayerNorm), 3 times.
  O=gpurun_out/r03mm
  mkdir -p $O
  for m in 1 2 3; do
    timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -s -k "shipped_schedule_equals_reference" 2>&1 | grep -E "rel max err|passed|failed" | cut -c1-200 | tee -a $O/sched.txt
  done
  ;;
N)
  # round-3 GPU pass NN: multi-row LayerNorm backward with every input of every row read BEFORE the first store (-DFD_LN_BWD_MULTI_ROW -DFD_LN_BWD_PRELOAD): the schedule test.
  O=gpurun_out/r03nn
  mkdir -p $O
  export FAIRDIFF_LIB=$PWD/finetune_fair_diffusion_amd/libfairdiff_hip_bench.so
  timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -s -k "shipped_schedule_equals_reference" 2>&1 | grep -E "rel max err|passed|failed" | cut -c1-200 | tee -a $O/sched.txt
  ;;
O)
  # round-3 GPU pass OO: last check of the rebuilt libraries: smoke() and the kernel tests.
  O=gpurun_out/r03oo
  mkdir -p $O
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
  timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -2 | tee $O/pytest_kernels.log
  ;;
*) echo "usage: $0 <a..z, A..O>"; exit 2;;
esac
