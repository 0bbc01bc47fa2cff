#!/bin/bash
# kernel-trace of the skinny-GEMM microbenchmark, new kernel and (FD_GEMM_NOSKINNY) (FD_GEMM_SKINNY_RT=2) two row tiles per wave everywhere
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sk_new -o sk -- python3 $R/scratch/mb_skinny.py > $R/gpurun_out/sk_new.log 2>&1
export FD_GEMM_SKINNY_RT=2
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sk_rt2 -o sk -- python3 $R/scratch/mb_skinny.py > $R/gpurun_out/sk_rt2.log 2>&1
cd $R
python3 - <<'PY'
import sqlite3
for tag in ("sk_new", "sk_rt2"):
    c = sqlite3.connect(f"gpurun_out/{tag}/sk_results.db")
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
    rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    g = [r for r in rows if 'gemm' in r[0]]
    print(tag, len(g))
    for i in range(0, len(g), 55):      # 10 shapes x 55 launches
        blk = g[i:i + 55]; d = sorted(r[2] - r[1] for r in blk)
        print("  ", blk[0][0][:52], "grid", blk[0][3], "median %.1f us" % (d[len(d) // 2] / 1e3))
PY
