"""Summaries from rocprofv3's rocpd SQLite output (what `rocprofv3 ... -d DIR -o NAME` writes on this image).
   python tools/rocpd_summary.py stats  OUT.csv  RUN.db              # --kernel-trace --stats run
   python tools/rocpd_summary.py pmc    OUT.csv  PASS1.db PASS2.db…  # one --pmc pass per database
FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE tallies a wide (16 B/lane) streaming read at half its
bytes (MI355X_MICROARCH.md, HBM / rocprofv3 section), hence the x2-corrected column."""
import csv, sqlite3, sys, collections

mode, out, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
if mode == "stats":
    c = sqlite3.connect(dbs[0])
    # one row per (kernel, grid): the same GEMM instantiation serves launches of different M (temporal vs strided blocks)
    rows = c.execute("select name, grid_x, grid_y, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name, grid_x, grid_y order by 5 desc").fetchall()
    tot = sum(r[4] for r in rows)
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "GridX", "GridY", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, gx, gy, k, t, a, lo, hi in rows:
            w.writerow([n, gx, gy, k, int(t), round(a, 1), round(100.0 * t / tot, 3), lo, hi])
    print("wrote", out, len(rows), "kernels, total", round(tot / 1e6, 3), "ms")
else:
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d in dbs:
        c = sqlite3.connect(d)
        for name, gx, gy, gs, ctr, val in c.execute("select kernel_name, grid_size_x, grid_size_y, grid_size, counter_name, value from counters_collection"):
            a = acc[(name, gs, gx, gy)][ctr]; a[0] += float(val); a[1] += 1
    cols = ["FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"]
    extra = sorted({x for cc in acc.values() for x in cc} - set(cols))      # any other counter of the run: plain per-launch averages
    rows = []
    for k, cc in acc.items():
        n = max(v[1] for v in cc.values())
        avg = {x: (cc[x][0] / cc[x][1] if x in cc and cc[x][1] else None) for x in cols}
        # SQ_VALU_MFMA_BUSY_CYCLES sums busy cycles over the 1024 SIMDs, GRBM_GUI_ACTIVE sums active cycles over the 8 XCDs
        frac = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0) if avg["SQ_VALU_MFMA_BUSY_CYCLES"] is not None and avg["GRBM_GUI_ACTIVE"] else None
        rows.append((-(avg["GRBM_GUI_ACTIVE"] or 0) * n - 1e-9 * sum(v[0] for v in cc.values()), [k[0], k[1], k[2], k[3], n, avg["FETCH_SIZE"], None if avg["FETCH_SIZE"] is None else avg["FETCH_SIZE"] * 1024 * 2,
                     avg["WRITE_SIZE"], None if avg["WRITE_SIZE"] is None else avg["WRITE_SIZE"] * 1024,
                     avg["SQ_VALU_MFMA_BUSY_CYCLES"], avg["SQ_BUSY_CU_CYCLES"], avg["GRBM_GUI_ACTIVE"], frac] +
                     [(cc[x][0] / cc[x][1] if x in cc and cc[x][1] else None) for x in extra]))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "grid_size", "grid_x", "grid_y", "launches", "FETCH_SIZE_KB_avg_raw", "HBM_read_bytes_avg_x2_gfx950_corrected", "WRITE_SIZE_KB_avg",
                    "HBM_write_bytes_avg", "SQ_VALU_MFMA_BUSY_CYCLES_avg", "SQ_BUSY_CU_CYCLES_avg", "GRBM_GUI_ACTIVE_avg", "mfma_pipe_busy_frac"] + [x + "_avg" for x in extra])
        for _, r in sorted(rows, key=lambda t: t[0]):
            w.writerow(["" if v is None else (round(v, 4) if isinstance(v, float) else v) for v in r])
    print("wrote", out, len(acc), "kernels")
