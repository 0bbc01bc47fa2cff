"""rocprofv3 counter CSVs of scratch/prof_pmc_r02.sh -> profiles/r02_pmc_traffic.json (per kernel name as bench.py's roofline spells it)."""
import csv, glob, json, os, re, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def load(pat, counter):
    rows = collections.defaultdict(list)
    for f in glob.glob(os.path.join(R, pat), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return rows
fetch = load("gpurun_out/pmc_r02_fetch/**/*counter_collection.csv", "FETCH_SIZE")
write = load("gpurun_out/pmc_r02_write/**/*counter_collection.csv", "WRITE_SIZE")
# shapes in launch order of mb_pmc_r02.py, keyed by the GEMM kernel's grid size (threads): tiles * split * 1024 threads
shapes = [dict(level="32x32", B=16, H=32, C=640, split=1), dict(level="16x16", B=16, H=16, C=1280, split=2), dict(level="8x8", B=16, H=8, C=1280, split=8)]
out = {"corrections": "FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE (KB) as reported",
       "note": "FETCH_SIZE is tallied at the L2's fabric side: Infinity-Cache hits are included (upper bound on HBM bytes). Split-K launches: the fp32 "
               "partial slabs written by the GEMM and re-read by splitk_reduce_kernel are counted (both kernels' counters are added).",
       "kernels": {}}
name = "gemm_big_kernel<128, 320, 4, 4, 1>"
entry = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on scratch/mb_pmc_r02.py", "shapes": []}
for s in shapes:
    M, N, K = s["B"] * s["H"] ** 2, s["C"], 9 * s["C"]
    grid = ((M + 127) // 128) * (N // 320) * s["split"] * 1024
    def mean(d, key_pred):
        v = [x for (k, g), xs in d.items() if key_pred(k, g) for x in xs]
        return sum(v) / max(len(v), 1), len(v)
    f_g, n = mean(fetch, lambda k, g: "gemm_big_kernel<128, 320, 4, 4, 1>" in k.replace("true", "1") and g == grid)
    w_g, _ = mean(write, lambda k, g: "gemm_big_kernel<128, 320, 4, 4, 1>" in k.replace("true", "1") and g == grid)
    f_r = w_r = 0.0
    if s["split"] > 1:   # the reduce kernels that followed launches of this shape: identified by their position is not possible in the CSV;
        # use the per-shape reduce grid instead (min(2048, M*N/4/256) blocks of 256 threads)
        rg = min(2048, (M * N // 4 + 255) // 256) * 256
        f_r, _ = mean(fetch, lambda k, g: "splitk_reduce_kernel" in k and g == rg)
        w_r, _ = mean(write, lambda k, g: "splitk_reduce_kernel" in k and g == rg)
    alg = 2.0 * (M * s["C"] + N * K + M * N)
    hbm = 2 * 1024 * (f_g + f_r) + 1024 * (w_g + w_r)
    entry["shapes"].append(dict(s, M=M, N=N, K=K, launches=n, fetch_size_kb_raw=f_g + f_r, write_size_kb_raw=w_g + w_r, hbm_bytes=hbm,
                                algorithmic_bytes=alg, ratio=hbm / alg))
out["kernels"][name] = entry
json.dump(out, open(os.path.join(R, "profiles", "r02_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(entry["shapes"], indent=1))
