#!/bin/bash
# effective shader clock during the GEMM / conv and attention kernels: GRBM_GUI_ACTIVE / kernel duration
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_r02_clk_gemm -o g -- python3 $R/scratch/mb_pmc_r02.py > $R/gpurun_out/pmc_r02_clk_gemm.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_r02_clk_attn -o a -- $R/scratch/attn_fwd_experiment > $R/gpurun_out/pmc_r02_clk_attn.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in ("gemm", "attn"):
    dur = {}
    for f in glob.glob(f"gpurun_out/pmc_r02_clk_{tag}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_r02_clk_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
                ns, name = dur[r["Dispatch_Id"]]
                if ns > 20000:
                    acc[name[:64]].append((float(r["Counter_Value"]), ns))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(x[1] for x in kv[1]))[:8]:
        cyc = sum(x[0] for x in v); ns = sum(x[1] for x in v)
        print(f"{k:66s} launches {len(v):4d}  avg {ns / len(v) / 1e3:8.1f} us  GRBM_GUI_ACTIVE/ns = {cyc / ns:6.3f} (GHz if the counter ticks once per shader clock; may be summed over 8 XCDs: /8 = {cyc / ns / 8:5.3f})")
PY
