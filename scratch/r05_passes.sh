#!/bin/bash
# every gpurun pass of round 5 is one case branch:  bash scratch/r05_passes.sh <letter>
mkdir -p gpurun_out
P=$PWD/finetune_fair_diffusion_amd
R=$PWD
B="python bench.py --no_cpu_baseline --no_roofline"
bench_table() {   # bench_table <glob>: value, ms/step, median, phases of every bench line that matches
python - "$1" <<'PY'
import json, glob, statistics, sys
for f in sorted(glob.glob(sys.argv[1])):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); h = d['config']['host_ms_per_step']
        print(f, round(d['value'], 3), round(d['ms_per_step'], 1), 'median', round(statistics.median(h), 1), d['config']['phase_ms'])
    except Exception as e:
        print(f, 'ERR', e)
PY
}
case "$1" in
a)  # evidence first: per-dispatch kernel trace of the shipped three-stream step (grid sizes, concurrency timeline), per-shape single-stream
    # times (--dump_shapes), then the in-situ PMC passes: SQ wave-cycle breakdown, FETCH_SIZE, WRITE_SIZE (separate passes, kernel-trace only)
    O=gpurun_out/r05a; mkdir -p $O
    cd /tmp && export TMPDIR=/tmp
    timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_r05a -o r05a -- python3 $R/bench.py --steps 2 --warmup 2 --no_cpu_baseline --no_roofline > $R/$O/bench_trace.json 2> $R/$O/bench_trace.err
    cd $R
    DB=$(find /tmp/prof_r05a -name "*.db" | head -1)
    python scratch/profsum.py $DB $O/kernel_stats.csv 30 > $O/kernel_stats_top.txt
    python scratch/r05_trace_rows.py $DB $O > $O/trace_rows.txt 2>&1; tail -45 $O/trace_rows.txt | cut -c1-200
    timeout 900 python bench.py --steps 4 --warmup 2 --no_cpu_baseline --dump_shapes $O/gemm_shapes.csv > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
    KR='gemm|attn|gn_|layernorm|geglu|splitk'
    cd /tmp
    C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES"
    timeout 1200 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "$KR" --output-format csv -d /tmp/pmc_r05a_sq -o s -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/pmc_sq.log 2>&1
    timeout 1200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$KR" --output-format csv -d /tmp/pmc_r05a_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/pmc_fetch.log 2>&1
    timeout 1200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$KR" --output-format csv -d /tmp/pmc_r05a_write -o w -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/pmc_write.log 2>&1
    cd $R
    python scratch/r05_pmc_summary.py /tmp/pmc_r05a_sq $O/pmc_sq_in_situ.csv 30 > $O/pmc_sq_top.txt 2>&1; cut -c1-230 $O/pmc_sq_top.txt
    python scratch/r05_pmc_summary.py /tmp/pmc_r05a_fetch $O/pmc_fetch_in_situ.csv 30 > $O/pmc_fetch_top.txt 2>&1; head -12 $O/pmc_fetch_top.txt | cut -c1-200
    python scratch/r05_pmc_summary.py /tmp/pmc_r05a_write $O/pmc_write_in_situ.csv 30 > $O/pmc_write_top.txt 2>&1; head -12 $O/pmc_write_top.txt | cut -c1-200
    ;;
b)  # register-B GEMMs (csrc/gemm_rb.hip): bit-exactness against the round-4 kernels on awkward shapes, then the A/B on the step's dense shapes with COLD A operands
    L=$P/libfairdiff_hip_bench.so
    MB_RB_NOASSERT=1 FAIRDIFF_LIB=$L timeout 900 python scratch/mb_rb.py $2 2>&1 | grep -v amdgpu.ids > gpurun_out/r05b_mb_rb.txt; cat gpurun_out/r05b_mb_rb.txt
    ;;
c)  FAIRDIFF_LIB=$P/libfairdiff_hip_bench.so timeout 600 python scratch/dbg_rb.py 2>&1 | grep -v amdgpu.ids | cut -c1-220 ;;
d)  # round-5 goldens on the GPU box's host CPU (the oracle's autograd graph at SD-v1.5 size needs more RAM than the build container has)
    O=gpurun_out/r05d; mkdir -p $O
    free -g | head -2; nproc
    for w in loss_seeds smooth_te cfg0_b8; do
      timeout 2400 python tests/golden/make_oracle_step_golden.py $w > $O/make_$w.log 2>&1; tail -3 $O/make_$w.log
    done
    cp tests/golden/oracle_sd15_smooth_head_te_lora_b2_s4.npz tests/golden/oracle_sd15_cfg0_b8_s4_te_lora.npz tests/golden/oracle_sd15_loss_seeds_b2_s2.npz $O/ 2>/dev/null; ls -la $O
    ;;
e)  # multi-rank hardening: the eight-rank launch ten times (first attempts only, tracebacks kept), then the S = 50 soak with a resident RCCL communicator
    O=gpurun_out/r05e; mkdir -p $O
    for i in 1 2 3 4 5 6 7 8 9 10; do
      timeout 900 python -m pytest tests/test_two_rank_gpu.py -q -s -k eight > $O/eight_$i.log 2>&1; rc=$?
      echo "run $i: rc=$rc $(tail -1 $O/eight_$i.log)" | tee -a $O/eight_summary.txt
      [ $rc -ne 0 ] && grep -v "amdgpu.ids\|socket.cpp\|Gloo" $O/eight_$i.log | tail -60 >> $O/eight_failures.txt
    done
    timeout 1500 python scratch/soak_s50.py 12 collectives 2>&1 | grep -v amdgpu.ids > $O/soak_s50_collectives.txt; tail -3 $O/soak_s50_collectives.txt | cut -c1-400
    ;;
f)  # ADVICE r4 fixes (attention D-fold pieces, first-tile rescale, graph keys) + the new tight goldens
    O=gpurun_out/r05f; mkdir -p $O
    timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" > $O/attention_tests.log 2>&1; tail -3 $O/attention_tests.log
    timeout 900 python -m pytest tests/test_engine_gpu.py -q -x -s -k "graphed or prefetch" > $O/graph_tests.log 2>&1; tail -3 $O/graph_tests.log
    timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -s -k "smooth_head_text_encoder or cfg0_eight or cfg0_step" > $O/golden_tests.log 2>&1; grep -i "cosine\|loss_fair\|passed\|failed\|Error\|assert" $O/golden_tests.log | tail -30
    ;;
g)  # is the norm ratio of the text-encoder-LoRA gradient (1.036 smooth head, 0.90 ReLU B = 8) rounding noise or a systematic factor?  Same tests under rounding-only switches
    O=gpurun_out/r05g; mkdir -p $O
    for v in "" "FD_NO_PRESCALED_Q=1" "FD_NO_GN_STATS=1" "FD_HOST_SCALES=1" "FD_ATOMIC_DKDV=1"; do
      n=$(echo "$v" | tr '=' '_'); [ -z "$n" ] && n=default
      env $v timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -s -k "smooth_head_text_encoder or cfg0_eight" > $O/golden_$n.log 2>&1
      echo "## $n"; grep -i "cosine\|loss_fair\|passed\|failed" $O/golden_$n.log | cut -c1-200
    done | tee $O/summary.txt
    ;;
h)  # the packed-fp32 hazard under other queue / stream counts and with zero-initialised registers.  SLP build of the CURRENT tree
    # (55 194 packed-fp32 VALU instructions), every layernorm_bwd of the concurrent backward executed twice (scratch/diag_hazard3.py), 4 steps per arm
    O=gpurun_out/r05h; mkdir -p $O
    run() { # name, env...
      n=$1; shift
      env FD_ALLOW_PACKED_FP32=1 "$@" timeout 600 python scratch/diag_hazard3.py 4 2>&1 | grep -v amdgpu.ids > $O/$n.txt
      echo "## $n: $(grep 'pairs executed' $O/$n.txt) | $(grep 'pairs differed' $O/$n.txt | tr '\n' ';')"
    }
    run slp_q8_s3 FAIRDIFF_LIB=$P/libfairdiff_hip_slp.so
    run slp_q4_s3 FAIRDIFF_LIB=$P/libfairdiff_hip_slp.so GPU_MAX_HW_QUEUES=4
    run slp_q1_s3 FAIRDIFF_LIB=$P/libfairdiff_hip_slp.so GPU_MAX_HW_QUEUES=1
    run slp_q8_s2 FAIRDIFF_LIB=$P/libfairdiff_hip_slp.so FD_BWD_STREAMS=2
    run slp_q8_s1 FAIRDIFF_LIB=$P/libfairdiff_hip_slp.so FD_NO_CONCURRENT_BWD=1
    run slp_zero_init_q8_s3 FAIRDIFF_LIB=$P/libfairdiff_hip_slp_zi.so
    run shipped_q8_s3 FD_DUMMY=1
    run slp_q8_s3_again FAIRDIFF_LIB=$P/libfairdiff_hip_slp.so
    ;;
j)  # what bounds the GEMM / conv kernels on COLD operands: L2 hit rates and request volumes, fabric requests, L1 -> L2 read latency, TA / TD stalls
    O=gpurun_out/r05j; mkdir -p $O
    cd /tmp && export TMPDIR=/tmp
    i=0
    for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_BUSY_avr" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
      i=$((i+1))
      timeout 600 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "gemm" --output-format csv -d /tmp/pmc_r05j_$i -o p -- python3 $R/scratch/mb_pmc_r05.py > $R/$O/pass_$i.log 2>&1
      python3 $R/scratch/r05_pmc_summary.py /tmp/pmc_r05j_$i $R/$O/pmc_pass_$i.csv 0 > /dev/null 2>&1
    done
    cd $R; cp gpurun_out/pmc_r05_manifest.json $O/ 2>/dev/null; ls -la $O
    ;;
k)  timeout 1200 python scratch/mb_pp_ablate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05k_pp_ablation.txt ;;
z)  # final validation of the tree: full GPU suite, smoke, the driver's default bench line, a 20-step line, the fused cross-attention A/B, a kernel trace
    O=gpurun_out/r05z; mkdir -p $O
    python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
    python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
    python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-250 $O/bench_default.json
    for i in 1 2; do
      for v in "FD_NO_FUSED_CROSS=1" "FD_NOTHING=1"; do
        n=$(echo "$v" | tr '=;, ' '____')
        env $v $B --steps 6 --warmup 2 > $O/ab_${n}_$i.json 2> $O/ab_${n}_$i.err || tail -3 $O/ab_${n}_$i.err
      done
    done
    bench_table "$O/ab_*.json" | tee $O/ab_summary.txt
    python bench.py --steps 20 --warmup 5 --cpu_baseline_bounded > $O/bench_steps20.json 2> $O/bench_steps20.err; cut -c1-250 $O/bench_steps20.json
    cd /tmp && export TMPDIR=/tmp
    timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_r05z -o r05z -- python3 $R/bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline > $R/$O/bench_trace.json 2> $R/$O/bench_trace.err
    cd $R
    DB=$(find /tmp/prof_r05z -name "*.db" | head -1)
    python scratch/profsum.py $DB $O/kernel_stats.csv 30 > $O/kernel_stats_top.txt; head -12 $O/kernel_stats_top.txt | cut -c1-200
    ;;
x)  # schedule knobs re-measured on the final tree: backward streams, R2 prefetch depth
    O=gpurun_out/r05x2; mkdir -p $O
    for i in 1 2; do
      for v in "FD_NOTHING=1" "FD_BWD_STREAMS=2" "FD_BWD_STREAMS=4" "FD_R2_PREFETCH_STEPS=6" "FD_R2_PREFETCH_STEPS=10" "FD_R2_PREFETCH_STEPS=12"; do
        n=$(echo "$v" | tr '=;, ' '____')
        env $v $B --steps 6 --warmup 2 > $O/${n}_$i.json 2> $O/${n}_$i.err || tail -3 $O/${n}_$i.err
      done
    done
    bench_table "$O/*.json" | tee $O/summary.txt
    ;;
w)  # GEGLU backward fused into the FF2 data-gradient GEMM's epilogue: parity, then the whole step
    O=gpurun_out/r05w2; mkdir -p $O
    python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -s -k "geglu" 2>&1 | tail -3
    for i in 1 2 3; do
      for v in "FD_NO_FUSED_GEGLU_BWD=1" "FD_NOTHING=1"; do
        n=$(echo "$v" | tr '=;, ' '____')
        env $v $B --steps 6 --warmup 2 > $O/${n}_$i.json 2> $O/${n}_$i.err || tail -3 $O/${n}_$i.err
      done
    done
    bench_table "$O/*.json" | tee $O/summary.txt
    ;;
v)  # dense short-K GEMMs as two co-resident 128x320 workgroups per CU (gemm_l2_kernel, bench-hooks library): isolated, then the whole step
    O=gpurun_out/r05v2; mkdir -p $O
    L=$P/libfairdiff_hip_bench.so
    FAIRDIFF_LIB=$L python scratch/mb_l2.py 2>&1 | grep -v amdgpu | tee $O/mb_default.txt
    FAIRDIFF_LIB=$L FD_GEMM_L2=1 python scratch/mb_l2.py 2>&1 | grep -v amdgpu | tee $O/mb_l2.txt
    for i in 1 2; do
      for v in "FD_NOTHING=1" "FD_GEMM_L2=1" "FD_GEMM_L2=1 FD_GEMM_L2_MAXK=640"; do
        n=$(echo "$v" | tr '=;, ' '____')
        env FAIRDIFF_LIB=$L $v $B --steps 6 --warmup 2 > $O/${n}_$i.json 2> $O/${n}_$i.err || tail -3 $O/${n}_$i.err
      done
    done
    bench_table "$O/*.json" | tee $O/summary.txt
    ;;
u)  # fused cross-attention with LoRA slabs + recording (R1 / R3 too): whole-step A/B, alternating arms
    O=gpurun_out/r05u; mkdir -p $O
    for i in 1 2 3; do
      for v in "FD_NO_FUSED_CROSS=1" "FD_NO_FUSED_CROSS_TRAIN=1" "FD_NOTHING=1"; do
        n=$(echo "$v" | tr '=;, ' '____')
        env $v $B --steps 6 --warmup 2 > $O/${n}_$i.json 2> $O/${n}_$i.err || tail -3 $O/${n}_$i.err
      done
    done
    bench_table "$O/*.json" | tee $O/summary.txt
    ;;
r)  # is the step bound by the host's enqueue rate?  a busy-wait in front of every C-ABI call (FD_HOST_SPIN_US), whole step
    O=gpurun_out/r05r; mkdir -p $O
    for i in 1 2; do
      for v in "FD_HOST_SPIN_US=0" "FD_HOST_SPIN_US=1" "FD_HOST_SPIN_US=2" "FD_HOST_SPIN_US=4"; do
        n=$(echo "$v" | tr '=;, ' '____')
        env $v $B --steps 6 --warmup 2 > $O/${n}_$i.json 2> $O/${n}_$i.err || tail -3 $O/${n}_$i.err
      done
    done
    bench_table "$O/*.json" | tee $O/summary.txt
    python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05r/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['config'].get('host_ms_between_phase_marks'))
PY
    ;;
q)  # would hipBLASLt's full-chip persistent tiles pay inside the step?  plain GEMMs of the medium-M levels routed through torch (measurement switch FD_BLASLT)
    O=gpurun_out/r05q; mkdir -p $O
    for i in 1 2; do
      for v in "FD_NOTHING=1" "FD_BLASLT=1" "FD_BLASLT=1 FD_BLASLT_MAXM=65536 FD_BLASLT_MINN=320" "FD_BLASLT=1 FD_BLASLT_MAXM=4096"; do
        n=$(echo "$v" | tr '=;, ' '____')
        env $v $B --steps 6 --warmup 2 > $O/${n}_$i.json 2> $O/${n}_$i.err || tail -3 $O/${n}_$i.err
      done
    done
    bench_table "$O/*.json" | tee $O/summary.txt
    ;;
p)  # calibration against hipBLASLt (torch): which kernels it picks on the step's dense shapes, and the tile-policy thresholds re-measured inside the step
    O=gpurun_out/r05p; mkdir -p $O
    cd /tmp && export TMPDIR=/tmp
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_r05p -o p -- python3 $R/scratch/mb_blaslt_names.py > $R/$O/names.log 2>&1
    cd $R
    find /tmp/prof_r05p -name "*kernel_stats*" | head -2
    python - <<'PY' > $O/blaslt_kernels.txt
import glob, csv
for f in glob.glob('/tmp/prof_r05p/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(r.get('Name', '')[:260], r.get('Calls'), r.get('AverageNs'))
PY
    cat $O/blaslt_kernels.txt | cut -c1-300
    L=$P/libfairdiff_hip_bench.so
    for i in 1 2; do
      for v in "FD_NOTHING=1" "FD_GEMM_T256=200 FD_GEMM_T128=160" "FD_GEMM_T256=160 FD_GEMM_T128=128" "FD_GEMM_T256=257 FD_GEMM_T128=200"; do
        n=$(echo "$v" | tr '=;, ' '____')
        env FAIRDIFF_LIB=$L $v $B --steps 6 --warmup 2 > $O/tile_${n}_$i.json 2> $O/tile_${n}_$i.err || tail -3 $O/tile_${n}_$i.err
      done
    done
    bench_table "$O/tile_*.json" | tee $O/tile_summary.txt
    ;;
n)  # fused cross-attention sub-block: parity, isolated timing, whole-step A/B (alternating arms)
    O=gpurun_out/r05n; mkdir -p $O
    python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -s -k "cross_attn_block" > $O/test_cross.log 2>&1; tail -3 $O/test_cross.log
    python scratch/mb_cross.py 2>&1 | tee $O/mb_cross.txt
    for i in 1 2; do
      for v in "FD_NO_FUSED_CROSS=1" "FD_FUSED_CROSS_C=320" "FD_FUSED_CROSS_C=320,640" "FD_FUSED_CROSS_C=320,640,1280"; do
        n=$(echo "$v" | tr '=;, ' '____')
        env $v $B --steps 6 --warmup 2 > $O/${n}_$i.json 2> $O/${n}_$i.err || tail -3 $O/${n}_$i.err
      done
    done
    bench_table "$O/*.json" | tee $O/summary.txt
    ;;
m)  # (needs scratch/r05_cu_mask_streams.patch) follow-up of pass l: only the PREFETCH of the next step's frozen rollout is CU-confined (its own stream "r2p"), everything else unmasked
    O=gpurun_out/r05m; mkdir -p $O
    for i in 1 2; do
      for v in "FD_NOTHING=1" "FD_CU_MASK=r2p=128-255" "FD_CU_MASK=r2p=192-255" "FD_CU_MASK=r2p=224-255" "FD_CU_MASK=r2p=128-255 FD_R2_PREFETCH_STEPS=12" "FD_CU_MASK=r2p=192-255 FD_R2_PREFETCH_STEPS=12" "FD_CU_MASK=r2p=0-255"; do
        n=$(echo "$v" | tr '=; ' '___'); [ "$v" = "FD_NOTHING=1" ] && n=default
        env $v $B --steps 6 --warmup 2 > $O/${n}_$i.json 2> $O/${n}_$i.err || tail -3 $O/${n}_$i.err
      done
    done
    bench_table "$O/*.json" | tee $O/summary.txt
    ;;
l)  # (needs scratch/r05_cu_mask_streams.patch applied to step.py) VERDICT r4 item 1d: CU-masked streams (hipExtStreamCreateWithCUMask) for R1 || R2 and the backward streams, whole-step A/B, alternating arms
    O=gpurun_out/r05l; mkdir -p $O
    for i in 1 2; do
      for v in "FD_NOTHING=1" "FD_CU_MASK=r2=128-255;main=0-127" ${R05L_ALL:+"FD_CU_MASK=r2=128-255" "FD_CU_MASK=r2=128-255;1=0-127;2=128-255" "FD_CU_MASK=r2=192-255"}; do
        n=$(echo "$v" | tr '=;' '__'); [ "$v" = "FD_NOTHING=1" ] && n=default
        env "$v" $B --steps 6 --warmup 2 > $O/${n}_$i.json 2> $O/${n}_$i.err || tail -3 $O/${n}_$i.err
      done
    done
    bench_table "$O/*.json" | tee $O/summary.txt
    ;;
*) echo "unknown pass $1";;
esac
