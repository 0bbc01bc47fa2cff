#!/bin/bash
# every gpurun pass of round 6 is one case branch:  bash scratch/r06_passes.sh <letter>
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
a)  # what bounds the main loops -- LDS-array and matrix-pipe busy shares per kernel, isolated (the step's shapes, cold operands) and in situ
    O=gpurun_out/r06a; mkdir -p $O
    $B --steps 6 --warmup 2 > $O/bench_start_of_round.json 2> $O/bench_start_of_round.err; cut -c1-200 $O/bench_start_of_round.json
    cd /tmp && export TMPDIR=/tmp
    i=0
    for C in "SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      timeout 600 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "gemm" --output-format csv -d /tmp/pmc_r06a_iso_$i -o p -- python3 $R/scratch/mb_pmc_r05.py > $R/$O/iso_pass_$i.log 2>&1
      timeout 1200 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "gemm" --output-format csv -d /tmp/pmc_r06a_situ_$i -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/situ_pass_$i.log 2>&1
    done
    cd $R
    python scratch/r06_pmc_lds_summary.py $O/pmc_lds_mainloop_isolated.txt /tmp/pmc_r06a_iso_1 /tmp/pmc_r06a_iso_2 | cut -c1-330
    python scratch/r06_pmc_lds_summary.py $O/pmc_lds_mainloop_in_situ.txt /tmp/pmc_r06a_situ_1 /tmp/pmc_r06a_situ_2 | head -30 | cut -c1-330
    cp gpurun_out/pmc_r05_manifest.json $O/ 2>/dev/null
    ;;
b)  # the halo-staged convolution: parity (every tile geometry, borders, epilogues), bit-equality with the per-tap kernel and the isolated A/B on the step's shapes
    O=gpurun_out/r06b; mkdir -p $O
    timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -s -k "conv3x3 or statistics_epilogue" > $O/conv_tests.log 2>&1; tail -5 $O/conv_tests.log; grep "conv halo" $O/conv_tests.log | head -30
    timeout 900 python scratch/mb_halo.py 2>&1 | grep -v amdgpu.ids | tee $O/mb_halo.txt
    ;;
c)  # whole step with the halo-staged convolutions (same box: the bench-hooks library with FD_CONV_HALO=0 / 1 alternating, then the product library)
    O=gpurun_out/r06c; mkdir -p $O
    for i in 1 2; do
      for v in "FD_CONV_HALO=0" "FD_CONV_HALO=1"; do
        n=$(echo "$v" | tr '=;, ' '____')
        env FAIRDIFF_LIB=$P/libfairdiff_hip_bench.so $v $B --steps 6 --warmup 2 > $O/${n}_$i.json 2> $O/${n}_$i.err || tail -3 $O/${n}_$i.err
      done
    done
    $B --steps 8 --warmup 2 > $O/product.json 2> $O/product.err || tail -3 $O/product.err
    bench_table "$O/*.json" | tee $O/summary.txt
    ;;
d)  # dense GEMMs on the ping-pong kernel now that its operands go through buffer descriptors: parity tests, then the isolated A/B against the shipped policy
    O=gpurun_out/r06d; mkdir -p $O
    timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" > $O/gemm_tests.log 2>&1; tail -3 $O/gemm_tests.log
    timeout 900 python scratch/mb_pp_dense.py 2>&1 | grep -v amdgpu.ids | tee $O/mb_pp_dense.txt
    ;;
e)  # chip fill under the shipped schedule, from per-workgroup (CU, start, end) records of the instrumented bench-hooks kernels
    O=gpurun_out/r06e; mkdir -p $O
    timeout 1200 python scratch/wg_fill.py $O/wg_fill_shipped_schedule.txt 3 2>&1 | grep -v amdgpu.ids | tail -60 | cut -c1-400
    ;;
f)  # kernel-time ranking of the current tree (rocprofv3 --kernel-trace of a bench run -> per-kernel stats) + the corrected LDS / MFMA counter summary (pass a's two counter sets)
    O=gpurun_out/r06f; mkdir -p $O
    cd /tmp && export TMPDIR=/tmp
    timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_r06f -o r06f -- python3 $R/bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline > $R/$O/bench_trace.json 2> $R/$O/bench_trace.err
    cd $R
    DB=$(find /tmp/prof_r06f -name "*.db" | head -1)
    python scratch/profsum.py $DB $O/kernel_stats.csv 45 > $O/kernel_stats_top.txt; head -50 $O/kernel_stats_top.txt | cut -c1-200
    cd /tmp
    i=0
    for C in "SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      timeout 600 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "gemm|conv_halo" --output-format csv -d /tmp/pmc_r06f_iso_$i -o p -- python3 $R/scratch/mb_pmc_r05.py > $R/$O/iso_pass_$i.log 2>&1
      timeout 1200 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "gemm|conv_halo" --output-format csv -d /tmp/pmc_r06f_situ_$i -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/situ_pass_$i.log 2>&1
    done
    cd $R
    python scratch/r06_pmc_lds_summary.py $O/pmc_lds_mainloop_isolated.txt /tmp/pmc_r06f_iso_1 /tmp/pmc_r06f_iso_2 | cut -c1-300
    python scratch/r06_pmc_lds_summary.py $O/pmc_lds_mainloop_in_situ.txt /tmp/pmc_r06f_situ_1 /tmp/pmc_r06f_situ_2 | head -24 | cut -c1-300
    ;;
g)  # full GPU suite + smoke on the current tree (durations kept)
    O=gpurun_out/r06g; mkdir -p $O
    python -m pytest tests -m gpu -x -q -s --durations=25 > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -40 $O/pytest_gpu.log | cut -c1-200
    python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
    ;;
h)  # halo convolution knobs inside the step (bench-hooks library): its own tile threshold (64: the 16^2 level on 256x320 tiles = 64 workgroups; 200: the 32^2 level on
    # 128x320 tiles = 256 workgroups), the nt cache policy on the A pieces
    O=gpurun_out/r06h; mkdir -p $O
    L=$P/libfairdiff_hip_bench.so
    for i in 1 2; do
      for v in "FD_NOTHING=1" "FD_CONV_T256=64" "FD_CONV_T256=200" "FAIRDIFF_LIB=$P/libfairdiff_hip_bench_nt.so"; do
        n=$(echo "$v" | sed 's/=.*bench_nt.so/_nt/' | tr '=;, /' '_____')
        env FAIRDIFF_LIB=$L $v $B --steps 6 --warmup 2 > $O/${n}_$i.json 2> $O/${n}_$i.err || tail -3 $O/${n}_$i.err
      done
    done
    bench_table "$O/*.json" | tee $O/summary.txt
    ;;
i)  # new tests of the round (loss bias over 64 seeds with the fp16-rounding denominator, exp-4 on eight ranks, lean recording, the fused-recording attention
    # backward) + the S = 50 exp-4 line with automatic lean recording against FD_LEAN_ACTIVATIONS=0 (round 3's schedule: whole timesteps recomputed)
    O=gpurun_out/r06i; mkdir -p $O
    timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -x -s -k "loss_fair_has_no_bias" > $O/loss_bias.log 2>&1; grep -i "loss_fair\|product\|fp16-rounded\|passed\|failed\|Error" $O/loss_bias.log | tail -12
    timeout 1500 python -m pytest tests/test_two_rank_gpu.py -q -x -s -k "exp4" > $O/exp4_eight.log 2>&1; grep -i "exp-4\|passed\|failed\|Error" $O/exp4_eight.log | tail -12
    timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_kernels_gpu.py -q -x -s -k "lean_recording or cross_attn_block_with_lora" > $O/lean_cross.log 2>&1; grep -i "lean\|fused recording\|passed\|failed\|Error" $O/lean_cross.log | tail -20
    for v in "FD_LEAN_ACTIVATIONS=0" "FD_NOTHING=1"; do
      n=$(echo "$v" | tr '=;, ' '____')
      env $v timeout 900 python bench.py --S 50 --experiment exp-4 --steps 3 --warmup 2 --no_cpu_baseline --no_roofline > $O/bench_exp4_s50_$n.json 2> $O/bench_exp4_s50_$n.err || tail -3 $O/bench_exp4_s50_$n.err
    done
    python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06i/bench_exp4_s50_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d["config"]
        print(f, round(d["value"], 3), "img/s", round(d["ms_per_step"], 1), "ms; kept", c["r3_timesteps_kept_in_hbm"], "of 50; ctx GB", c["r3_activation_gb_per_timestep"], "peak", c["peak_hbm_gb"], c["phase_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
    ;;
j)  # S = 50 exp-4 with automatic lean recording (after the pool hand-back at the mode switch), 4 timed steps
    O=gpurun_out/r06j; mkdir -p $O
    timeout 900 python bench.py --S 50 --experiment exp-4 --steps 4 --warmup 2 --no_cpu_baseline --no_roofline > $O/bench_exp4_s50_lean.json 2> $O/bench_exp4_s50_lean.err || tail -5 $O/bench_exp4_s50_lean.err
    python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r06j/bench_exp4_s50_lean.json").read().strip().splitlines()[-1]); c = d["config"]
    print(round(d["value"], 3), "img/s", round(d["ms_per_step"], 1), "ms; kept", c["r3_timesteps_kept_in_hbm"], "of 50; ctx GB", c["r3_activation_gb_per_timestep"], "peak", c["peak_hbm_gb"], c["phase_ms"], c["host_ms_per_step"])
except Exception as e:
    print("ERR", e)
PY
    ;;
k)  # the ViT GEMMs of the tail (M = 2112 tokens, N = 1280): 128x160 tiles instead of 64x64 (threshold of the 160-wide tile 160 -> 128 / 100), whole step
    O=gpurun_out/r06k; mkdir -p $O
    L=$P/libfairdiff_hip_bench.so
    for i in 1 2; do
      for v in "FD_NOTHING=1" "FD_GEMM_T160=128" "FD_GEMM_T160=100"; do
        n=$(echo "$v" | tr '=;, /' '_____')
        env FAIRDIFF_LIB=$L $v $B --steps 6 --warmup 2 > $O/${n}_$i.json 2> $O/${n}_$i.err || tail -3 $O/${n}_$i.err
      done
    done
    bench_table "$O/*.json" | tee $O/summary.txt
    ;;
z)  # final validation of the tree: full GPU suite, smoke, the driver's default bench line, a 20-step line, rocprofv3 kernel stats of the bench command,
    # the in-situ FETCH / WRITE passes that bench.py's roofline.traffic reads, the cfg0 B = 8 cosine without the fused cross block in the recording forward
    O=gpurun_out/r06z; mkdir -p $O
    python -m pytest tests -m gpu -x -q -s --durations=15 > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -22 $O/pytest_gpu.log | cut -c1-200
    python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
    python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-250 $O/bench_default.json
    python bench.py --steps 20 --warmup 5 --cpu_baseline_bounded > $O/bench_steps20.json 2> $O/bench_steps20.err; cut -c1-250 $O/bench_steps20.json
    env FD_NO_FUSED_CROSS_TRAIN=1 timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -s -k "cfg0_eight" > $O/cfg0_b8_no_fused_cross_train.log 2>&1; grep -i "cfg0 B=8\|passed\|failed" $O/cfg0_b8_no_fused_cross_train.log
    cd /tmp && export TMPDIR=/tmp
    timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_r06z -o r06z -- python3 $R/bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline > $R/$O/bench_trace.json 2> $R/$O/bench_trace.err
    cd $R
    DB=$(find /tmp/prof_r06z -name "*.db" | head -1)
    python scratch/profsum.py $DB $O/kernel_stats.csv 30 > $O/kernel_stats_top.txt; head -14 $O/kernel_stats_top.txt | cut -c1-200
    find /tmp/prof_r06z -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats.csv 2>/dev/null
    KR='gemm|conv_halo|attn|gn_|layernorm|geglu|splitk'
    cd /tmp
    timeout 1200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$KR" --output-format csv -d /tmp/pmc_r06z_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/pmc_fetch.log 2>&1
    timeout 1200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$KR" --output-format csv -d /tmp/pmc_r06z_write -o w -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/pmc_write.log 2>&1
    cd $R
    python scratch/r05_pmc_summary.py /tmp/pmc_r06z_fetch $O/pmc_fetch_in_situ.csv 12 > $O/pmc_fetch_top.txt 2>&1; head -8 $O/pmc_fetch_top.txt | cut -c1-200
    python scratch/r05_pmc_summary.py /tmp/pmc_r06z_write $O/pmc_write_in_situ.csv 12 > $O/pmc_write_top.txt 2>&1; head -8 $O/pmc_write_top.txt | cut -c1-200
    ;;
l)  # what would GroupNorm + SiLU applied to the staged A operand cost INSIDE the halo convolution?  The bench-hooks library against the same library with the
    # in-LDS transform pass compiled in (-DHALO_GN_PROBE, measurement only: the 128-row instantiations; the 256-row ones spill with it), isolated launches
    O=gpurun_out/r06l; mkdir -p $O
    FD_CONV_HALO=1 FAIRDIFF_LIB=$P/libfairdiff_hip_bench.so timeout 600 python scratch/mb_halo.py 2>&1 | grep -v amdgpu.ids > $O/halo_plain.txt
    FD_CONV_HALO=1 FAIRDIFF_LIB=$P/libfairdiff_hip_bench_gnp.so timeout 600 python scratch/mb_halo.py 2>&1 | grep -v amdgpu.ids > $O/halo_gn_probe.txt
    paste -d'\n' $O/halo_plain.txt $O/halo_gn_probe.txt | cut -c1-250
    timeout 600 python scratch/mb_gn.py 2>&1 | grep -v amdgpu.ids | tail -30 > $O/mb_gn.txt; cat $O/mb_gn.txt | cut -c1-200
    # the kernel trace of the SINGLE-STREAM schedule the roofline step of bench.py times (both stream overlaps off): its mean durations are the ones roofline.avg_launch_us must agree with
    cd /tmp && export TMPDIR=/tmp
    FD_NO_CONCURRENT_R2=1 FD_NO_CONCURRENT_BWD=1 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_r06l -o r06l -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/bench_trace_single_stream.json 2> $R/$O/bench_trace_single_stream.err
    cd $R
    DB=$(find /tmp/prof_r06l -name "*.db" | head -1)
    python scratch/profsum.py $DB $O/kernel_stats_single_stream.csv 12 > $O/kernel_stats_single_stream_top.txt; head -8 $O/kernel_stats_single_stream_top.txt | cut -c1-200
    ;;
m)  # S = 50 soak with the automatic lean recording (and with a resident RCCL communicator): memory flat, no allocator retries, all 50 timesteps kept
    O=gpurun_out/r06m; mkdir -p $O
    timeout 900 python scratch/soak_s50.py 14 2>&1 | grep -v amdgpu.ids > $O/soak_s50_lean.txt; tail -4 $O/soak_s50_lean.txt | cut -c1-400
    timeout 900 python scratch/soak_s50.py 14 collectives 2>&1 | grep -v amdgpu.ids > $O/soak_s50_lean_collectives.txt; tail -3 $O/soak_s50_lean_collectives.txt | cut -c1-400
    ;;
n)  # dense GEMMs on the 8-wave ping-pong kernel (buffer-descriptor operands, level in isolation) INSIDE the step: does leaving a third of the register file and
    # 15 KB of LDS to the other streams' small kernels change the in-situ picture?  FD_GEMM_PP bits: 2 = dense 256x320, 16 = dense 128x320
    O=gpurun_out/r06n; mkdir -p $O
    L=$P/libfairdiff_hip_bench.so
    for i in 1 2; do
      for v in "FD_GEMM_PP=45" "FD_GEMM_PP=47" "FD_GEMM_PP=63"; do
        n=$(echo "$v" | tr '=;, /' '_____')
        env FAIRDIFF_LIB=$L $v $B --steps 6 --warmup 2 > $O/${n}_$i.json 2> $O/${n}_$i.err || tail -3 $O/${n}_$i.err
      done
    done
    bench_table "$O/*.json" | tee $O/summary.txt
    ;;
o)  # the in-rollout switch to lean recording (first step of a run) + the 2 GiB reserve: the S = 50 test, the lean test, a short soak
    O=gpurun_out/r06o; mkdir -p $O
    timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_engine_gpu.py -q -x -s -k "s50_mixed or lean_recording" > $O/tests.log 2>&1; grep -i "S=50\|lean\|passed\|failed\|Error\|grad" $O/tests.log | tail -12 | cut -c1-250
    timeout 900 python scratch/soak_s50.py 8 2>&1 | grep -v amdgpu.ids > $O/soak_s50_lean.txt; grep "step\\\": 0,\|step\\\": 1,\|summary" $O/soak_s50_lean.txt | cut -c1-400
    ;;
p)  # where the single-stream tail goes: fine marks inside R3_loss_and_image_grad (CLIP / DINO forward, their backwards, the face branch, the classifier backward)
    O=gpurun_out/r06p; mkdir -p $O
    FD_FINE_MARKS=1 $B --steps 4 --warmup 2 > $O/fine_marks.json 2> $O/fine_marks.err || tail -3 $O/fine_marks.err
    python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06p/fine_marks.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"], 1), d["config"]["phase_ms"])
PY
    ;;
q)  # REJECTED (scratch/r06_concurrent_tail_rejected.patch, profiles/r06_step_ab_concurrent_tail_rejected.txt): concurrent loss tail (DINOv2 / face branch / classifier backward beside CLIP): tests, then the same-box A/B
    O=gpurun_out/r06q; mkdir -p $O
    timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py -q -x -k "regularis or face_realism or bit_reproducible or shipped_schedule or all_regularisers or detector_provider" > $O/tests.log 2>&1; tail -4 $O/tests.log | cut -c1-200
    for i in 1 2 3; do
      for v in "FD_NO_CONCURRENT_TAIL=1" "FD_NOTHING=1"; do
        n=$(echo "$v" | tr '=;, /' '_____')
        env $v $B --steps 6 --warmup 2 > $O/${n}_$i.json 2> $O/${n}_$i.err || tail -3 $O/${n}_$i.err
      done
    done
    bench_table "$O/*.json" | tee $O/summary.txt
    ;;
r)  # evidence passes on the final tree: SQ wave-cycle breakdown + LDS bank conflicts per kernel in situ (is the halo image's slot permutation conflict-free at every
    # tap shift?), and the exp-4 eight-rank test five times (first attempts only)
    O=gpurun_out/r06r; mkdir -p $O
    cd /tmp && export TMPDIR=/tmp
    KR='gemm|conv_halo|attn|gn_|layernorm|geglu|splitk'
    C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES"
    timeout 1200 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "$KR" --output-format csv -d /tmp/pmc_r06r_sq -o s -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline > $R/$O/pmc_sq.log 2>&1
    cd $R
    python scratch/r05_pmc_summary.py /tmp/pmc_r06r_sq $O/pmc_sq_in_situ.csv 30 > $O/pmc_sq_top.txt 2>&1; cut -c1-230 $O/pmc_sq_top.txt | head -34
    for i in 1 2 3 4 5; do
      timeout 900 python -m pytest tests/test_two_rank_gpu.py -q -s -k "exp4" > $O/exp4_$i.log 2>&1; rc=$?
      echo "run $i: rc=$rc $(grep 'eight ranks vs one rank' $O/exp4_$i.log | cut -c1-200) $(tail -1 $O/exp4_$i.log)" | tee -a $O/exp4_summary.txt
    done
    ;;
esac
