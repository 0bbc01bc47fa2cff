"""Join of the round-6 LDS / MFMA counter passes (separate rocprofv3 --pmc runs over the same launches) into per (kernel, workgroups) rows with the LDS-array
and matrix-pipe busy shares, normalised as profiles/r01_pmc_lds_conv256x320.txt did: SQ_LDS_IDX_ACTIVE / 256 CUs and SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs against
the dispatch's GRBM_GUI_ACTIVE cycles (/ 8: the counter sums the XCDs).      python scratch/r06_pmc_lds_summary.py <out.txt> <pass dir> [<pass dir> ...]"""
import csv, glob, os, re, sys, collections

out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return n[:100]


for src in sys.argv[2:]:
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            g = int(r.get("Grid_Size", 0) or 0)
            w = int(r.get("Workgroup_Size", 1) or 1)
            a = acc[(short(r["Kernel_Name"]), g // max(w, 1))][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
rows = []
for k, v in acc.items():
    m = {c: v[c][1] / v[c][0] for c in v}
    n = max(v[c][0] for c in v)
    rows.append((k, n, m))
rows.sort(key=lambda r: -r[1] * r[2].get("GRBM_GUI_ACTIVE", 0.0))
with open(out, "w") as f:
    f.write("per (kernel, workgroups): mean over the launches of each pass; LDS busy = SQ_LDS_IDX_ACTIVE / 256 / GRBM_GUI_ACTIVE, MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / GRBM_GUI_ACTIVE,\n"
            "LDS bytes/clk/CU = 256 x LDS busy at ds_read_b128 (guide: 256 B/clk/CU peak); wave-cycle counters are quad-cycles summed over waves\n")
    for (n, g), cnt, m in rows:
        gui = (m.get("GRBM_GUI_ACTIVE", 0.0) or 8.0) / 8.0        # the counter is summed over the 8 XCDs' GRBMs (a 230 us launch reads 3.7 M)
        occ = min(g, 256) / 256.0                                  # share of the CUs that hold a workgroup (one workgroup per CU in these kernels)
        wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        f.write("%-72s wgs %6d n %5d  cycles %9.0f  LDS-array busy %5.1f%% (occupied CUs %5.1f%%)  MFMA-pipe busy %5.1f%% (occupied CUs %5.1f%%)  LDS insts/CU %8.0f  active_inst_lds/wc %5.1f%%  vmem_inst_cycles/wc %5.1f%%  wait_inst_any/wc %5.1f%%  sq_busy_cyc %9.0f\n" % (
            n[:72], g, cnt, gui, 100 * m.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / gui, 100 * m.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / gui / occ, 100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / gui, 100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / gui / occ, m.get("SQ_INSTS_LDS", 0) / 256,
            100 * m.get("SQ_ACTIVE_INST_LDS", 0) / wc, 100 * m.get("SQ_INST_CYCLES_VMEM", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_BUSY_CYCLES", 0)))
print(open(out).read()[:6000])
