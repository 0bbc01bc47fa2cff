"""rocprofv3 counter CSVs of scratch/prof_pmc_r03.sh + the launch manifest of scratch/mb_pmc_r03.py -> profiles/r03_pmc_traffic.json, keyed by the
kernel name bench.py's roofline reports.  Launches are attributed to shapes by DISPATCH ORDER (several shapes share a grid size)."""
import csv, glob, json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
man = json.load(open(os.path.join(R, "gpurun_out", "pmc_r03_manifest.json")))
def load(pat, counter):
    rows = []
    for f in glob.glob(os.path.join(R, pat), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    return sorted(rows)
def per_shape(rows):
    gemm = [(d, v) for d, k, v in rows if "gemm_big_kernel" in k or "gemm_glds_kernel" in k or "gemm_pp_kernel" in k]
    red = sorted((d, v) for d, k, v in rows if "splitk_reduce_kernel" in k)
    out, i = [], 0
    for s in man:
        part = gemm[i:i + s["launches"]]
        i += s["launches"]
        g = sum(v for _, v in part) / len(part)
        r = 0.0
        if s["split"] > 1:      # the reduce kernel that directly follows each split GEMM
            vals = []
            for d, _ in part:
                nxt = [v for dd, v in red if dd == d + 1]
                vals.append(nxt[0] if nxt else 0.0)
            r = sum(vals) / len(vals)
        out.append((g, r, len(part)))
    assert i == len(gemm), (i, len(gemm))
    return out
F = per_shape(load("gpurun_out/pmc_r03_fetch/**/*counter_collection.csv", "FETCH_SIZE"))
W = per_shape(load("gpurun_out/pmc_r03_write/**/*counter_collection.csv", "WRITE_SIZE"))
out = {"corrections": "FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE (KB) as reported",
       "note": "FETCH_SIZE is tallied at the L2's fabric side: Infinity-Cache hits are included (upper bound on HBM bytes). Split-K launches: the fp32 "
               "partial slabs written by the GEMM and re-read by splitk_reduce_kernel are counted (both kernels' counters are added).",
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on scratch/mb_pmc_r03.py; launches attributed to shapes by dispatch order",
       "kernels": {}}
for s, (fg, fr, n), (wg, wr, _) in zip(man, F, W):
    hbm = 2 * 1024 * (fg + fr) + 1024 * (wg + wr)
    e = dict(s, fetch_size_kb_raw_gemm=fg, fetch_size_kb_raw_reduce=fr, write_size_kb_raw_gemm=wg, write_size_kb_raw_reduce=wr, hbm_bytes=hbm,
             ratio=hbm / s["algorithmic_bytes"])
    out["kernels"].setdefault(s["kernel"], {"shapes": []})["shapes"].append(e)
    print(s["kernel"], s["kind"], (s["M"], s["N"], s["K"]), "split", s["split"], "ratio %.2f" % e["ratio"], "MB %.1f" % (hbm / 1e6))
json.dump(out, open(os.path.join(R, "profiles", "r03_pmc_traffic.json"), "w"), indent=1)
