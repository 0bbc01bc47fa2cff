"""rocprofv3 rocpd database of a bench.py run -> (1) per-dispatch rows of the LAST timed step as a gzip CSV (name id, grid, workgroup, LDS, queue / stream, start, end),
(2) a per (kernel, grid) summary, (3) a concurrency timeline of that step: share of the step's wall time with k kernels in flight and the time-weighted sum of
min(workgroups in flight, 256) -- how full the chip is by launch geometry alone.      python scratch/r05_trace_rows.py <db> <outdir> [steps_in_trace]"""
import csv, gzip, os, re, sqlite3, sys, collections

db = sqlite3.connect(sys.argv[1])
out = sys.argv[2]
os.makedirs(out, exist_ok=True)
cur = db.cursor()
views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')").fetchall()]
open(os.path.join(out, "schema.txt"), "w").write("\n".join(views) + "\n" + "\n".join(str(r) for r in cur.execute("PRAGMA table_info(kernels)").fetchall()) + "\n")
cols = [r[1] for r in cur.execute("PRAGMA table_info(kernels)").fetchall()]


def pick(*names):
    for n in names:
        if n in cols:
            return n
    return None


c_grid = pick("grid_x", "grid_size_x", "grid_size")
c_wg = pick("workgroup_x", "workgroup_size_x", "workgroup_size")
c_lds = pick("lds_size", "lds_block_size", "group_segment_size")
c_q = pick("queue_id", "queue", "stream_id", "stream")
c_st = pick("stream_id", "stream")
sel = ["name", "start", "end"] + [c for c in (c_grid, c_wg, c_lds, c_q, c_st) if c]
rows = cur.execute("select %s from kernels order by start" % ", ".join(sel)).fetchall()
print("dispatches", len(rows), "columns", sel)


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return n[:110]


# the last step = everything after the last adamw_ema launch but one ... simpler: split the trace at the adamw_ema kernels (one per step)
ends = [i for i, r in enumerate(rows) if "adamw_ema" in r[0]]
print("optimiser launches (one per step):", len(ends))
if len(ends) >= 2:
    lo, hi = ends[-2] + 1, ends[-1] + 1
else:
    lo, hi = 0, len(rows)
step = rows[lo:hi]
t0 = min(r[1] for r in step)
t1 = max(r[2] for r in step)
print("last step: %d dispatches, %.1f ms wall (first start -> last end; includes the prefetched R2 of the next step that overlaps)" % (len(step), (t1 - t0) / 1e6))
names = {}
with gzip.open(os.path.join(out, "last_step_dispatches.csv.gz"), "wt") as f:
    w = csv.writer(f)
    w.writerow(["name_id", "start_us", "dur_us"] + sel[3:])
    for r in step:
        nid = names.setdefault(short(r[0]), len(names))
        w.writerow([nid, "%.2f" % ((r[1] - t0) / 1e3), "%.2f" % ((r[2] - r[1]) / 1e3)] + list(r[3:]))
with open(os.path.join(out, "last_step_names.csv"), "w") as f:
    for n, i in names.items():
        f.write("%d,\"%s\"\n" % (i, n))

# per (kernel, grid) summary
gi = sel.index(c_grid) if c_grid else None
wi = sel.index(c_wg) if c_wg else None
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    g = r[gi] if gi is not None else 0
    wgs = (g // r[wi]) if (gi is not None and wi is not None and r[wi]) else g
    k = (short(r[0]), wgs)
    agg[k][0] += 1
    agg[k][1] += (r[2] - r[1]) / 1e3
tot = sum(v[1] for v in agg.values())
with open(os.path.join(out, "last_step_by_kernel_and_grid.csv"), "w") as f:
    f.write("kernel,workgroups,calls,total_ms,avg_us,pct\n")
    for (n, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write("\"%s\",%d,%d,%.3f,%.2f,%.2f\n" % (n, g, c, t / 1e3, t / c, 100 * t / tot))
print("kernel time of the step %.1f ms" % (tot / 1e3))
for (n, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-90s wgs %6d calls %5d  %8.1f ms  avg %8.1f us  %5.1f%%" % (n[:90], g, c, t / 1e3, t / c, 100 * t / tot))

# concurrency timeline
ev = []
for r in step:
    g = r[gi] if gi is not None else 0
    wgs = (g // r[wi]) if (gi is not None and wi is not None and r[wi]) else 1
    ev.append((r[1], 1, wgs))
    ev.append((r[2], -1, -wgs))
ev.sort()
nk = 0
nwg = 0
last = ev[0][0]
hist = collections.defaultdict(float)
fill = 0.0
small = 0.0     # time with fewer than 256 workgroups in flight in total
for t, dk, dw in ev:
    dt = t - last
    hist[nk] += dt
    fill += dt * min(nwg, 256)
    if nwg < 256:
        small += dt
    nk += dk
    nwg += dw
    last = t
wall = t1 - t0
with open(os.path.join(out, "last_step_concurrency.txt"), "w") as f:
    f.write("wall %.1f ms, kernel-time sum %.1f ms (concurrency %.2f)\n" % (wall / 1e6, tot / 1e3, tot * 1e3 / wall))
    for k in sorted(hist):
        f.write("kernels in flight %2d: %6.1f ms (%5.1f %%)\n" % (k, hist[k] / 1e6, 100 * hist[k] / wall))
    f.write("time-weighted min(workgroups in flight, 256) / 256 = %.3f\n" % (fill / wall / 256))
    f.write("time with < 256 workgroups in flight: %.1f ms (%.1f %%)\n" % (small / 1e6, 100 * small / wall))
print(open(os.path.join(out, "last_step_concurrency.txt")).read())
