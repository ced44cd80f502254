"""Generic per-kernel averages of every counter found in rocprofv3 --pmc rocpd databases (one or more passes).
   python tools/rocpd_counters.py OUT.csv PASS1.db PASS2.db ...
One row per (kernel, grid_x, grid_y); counter columns are per-launch averages of the sum over all XCDs / SEs."""
import csv, sqlite3, sys, collections
out, dbs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
names = []
for d in dbs:
    c = sqlite3.connect(d)
    for name, gx, gy, ctr, val in c.execute("select kernel_name, grid_size_x, grid_size_y, counter_name, value from counters_collection"):
        a = acc[(name, gx, gy)][ctr]; a[0] += float(val); a[1] += 1
        if ctr not in names: names.append(ctr)
rows = []
for k, cc in acc.items():
    n = max(v[1] for v in cc.values())
    avg = {x: (cc[x][0] / cc[x][1] if x in cc and cc[x][1] else None) for x in names}
    rows.append((-(avg.get("SQ_WAVE_CYCLES") or avg.get("GRBM_GUI_ACTIVE") or 0) * n, [k[0], k[1], k[2], n] + [avg[x] for x in names]))
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "grid_x", "grid_y", "launches"] + names)
    for _, r in sorted(rows, key=lambda t: t[0]):
        w.writerow([("" if v is None else (round(v, 1) if isinstance(v, float) else v)) for v in r])
print("wrote", out, len(rows), "kernels,", len(names), "counters")
