"""rocprofv3 --pmc counter_collection CSVs of an IN-SITU bench step (real data flow, the kernels the step launches, in launch order; counter collection
serialises the dispatches) -> per (kernel, workgroups) means of every counter.     python scratch/r05_pmc_summary.py <dir with *counter_collection.csv> <out.csv> [top]
The SQ pass adds derived columns: share of wave-cycles parked (WAIT_ANY), issue-stalled (WAIT_INST_ANY), issuing (ACTIVE_INST_ANY), MFMA pipe busy per SIMD-cycle."""
import csv, glob, os, re, sys, collections

src, dst = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return n[:110]


acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
disp = collections.defaultdict(set)
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        g = int(r.get("Grid_Size", 0) or 0)
        w = int(r.get("Workgroup_Size", 1) or 1)
        k = (short(r["Kernel_Name"]), g // max(w, 1))
        a = acc[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
counters = sorted({c for v in acc.values() for c in v})
rows = []
for k, v in acc.items():
    m = {c: (v[c][1] / v[c][0] if c in v else 0.0) for c in counters}
    rows.append((k, len(disp[k]), m))
key = "SQ_WAVE_CYCLES" if "SQ_WAVE_CYCLES" in counters else counters[0]
rows.sort(key=lambda r: -r[1] * r[2].get(key, 0.0))
sq = "SQ_WAVE_CYCLES" in counters
with open(dst, "w") as f:
    hdr = ["kernel", "workgroups", "dispatches"] + counters
    if sq:
        hdr += ["wait_any_pct", "wait_inst_any_pct", "active_inst_any_pct", "wait_inst_lds_pct", "vmem_inst_cycles_pct", "mfma_busy_per_simd_cycle", "lds_conflict_per_wave_cycle"]
    f.write(",".join(hdr) + "\n")
    for (n, g), d, m in rows:
        line = ["\"%s\"" % n, str(g), str(d)] + ["%.6g" % m[c] for c in counters]
        if sq:
            wc = m["SQ_WAVE_CYCLES"] or 1.0
            # SQ_WAVE_CYCLES etc. count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_BUSY_CYCLES (if collected) per SE
            busy = m.get("SQ_BUSY_CU_CYCLES", 0.0) or m.get("SQ_BUSY_CYCLES", 0.0)
            line += ["%.1f" % (100 * m.get("SQ_WAIT_ANY", 0) / wc), "%.1f" % (100 * m.get("SQ_WAIT_INST_ANY", 0) / wc), "%.1f" % (100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc),
                     "%.1f" % (100 * m.get("SQ_WAIT_INST_LDS", 0) / wc), "%.1f" % (100 * m.get("SQ_INST_CYCLES_VMEM", 0) / wc),
                     "%.4f" % (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / busy if busy else 0.0), "%.4f" % (m.get("SQ_LDS_BANK_CONFLICT", 0) / wc)]
        f.write(",".join(line) + "\n")
print(open(dst).read().split("\n")[0])
for (n, g), d, m in rows[:top]:
    s = "%-80s wgs %6d n %5d " % (n[:80], g, d)
    if sq:
        wc = m["SQ_WAVE_CYCLES"] or 1.0
        s += "parked %5.1f%% stalled %5.1f%% issuing %5.1f%% ldsstall %4.1f%% mfma_busy_cyc %.3g conflicts/wc %.3f" % (
            100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_LDS", 0) / wc,
            m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), m.get("SQ_LDS_BANK_CONFLICT", 0) / wc)
    else:
        s += " ".join("%s=%.4g" % (c, m[c]) for c in counters)
    print(s)
