"""Per-stream occupancy of a rocprofv3 kernel trace (rocpd SQLite): how much of the wall time each queue / stream had a kernel
running, how much of it at least one had (union), and the largest gaps on the busiest stream.
   python tools/rocpd_timeline.py RUN.db [skip_fraction=0.5]      # analyses the second half of the trace by default"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print("columns:", cols)
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = c.execute(f"select start, end, {qcol}, name from kernels order by start").fetchall()
t0, t1 = rows[0][0], rows[-1][1]
lo = t0 + (t1 - t0) * skip
rows = [r for r in rows if r[0] >= lo]
wall = rows[-1][1] - rows[0][0]
per = collections.defaultdict(lambda: [0, 0])
for s, e, q, n in rows:
    per[q][0] += e - s; per[q][1] += 1
ev = sorted([(s, 1) for s, e, q, n in rows] + [(e, -1) for s, e, q, n in rows])
busy = 0; depth = 0; last = None; conc = collections.Counter()
for t, d in ev:
    if last is not None and depth > 0: busy += t - last
    if last is not None: conc[depth] += t - last
    depth += d; last = t
print(f"window {wall / 1e6:.3f} ms, {len(rows)} kernels; union busy {busy / 1e6:.3f} ms ({100 * busy / wall:.1f} %)")
for q, (b, k) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print(f"  {qcol} {q}: {k} kernels, busy {b / 1e6:.3f} ms ({100 * b / wall:.1f} %)")
print("time with k kernels in flight:", {k: f"{100 * v / wall:.1f} %" for k, v in sorted(conc.items())})
# gaps on the busiest stream: histogram and the kernels around the largest ones
main = max(per.items(), key=lambda kv: kv[1][0])[0]
mr = [r for r in rows if r[2] == main]
gaps = [(mr[i + 1][0] - mr[i][1], mr[i][3][:60], mr[i + 1][3][:60]) for i in range(len(mr) - 1)]
tot = sum(g for g, _, _ in gaps if g > 0)
hist = collections.Counter()
for g, _, _ in gaps:
    hist["<2us" if g < 2000 else "2-5us" if g < 5000 else "5-10us" if g < 10000 else "10-30us" if g < 30000 else ">30us"] += 1
print(f"stream {main}: {len(gaps)} gaps, total {tot / 1e6:.3f} ms; histogram {dict(hist)}")
big = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    if g >= 5000: big[(a, b)][0] += g; big[(a, b)][1] += 1
for (a, b), (g, k) in sorted(big.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {g / 1e3 / k:7.1f} us x {k:4d}  after {a}  ->  {b}")
# what the busiest stream spends its time on
top = collections.defaultdict(lambda: [0, 0])
for s, e, q, n in mr: top[n[:110]][0] += e - s; top[n[:110]][1] += 1
print(f"stream {main}: kernels by time")
for n, (b, k) in sorted(top.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"  {b / 1e6:8.3f} ms  {k:5d} x {b / 1e3 / k:7.1f} us  {n}")
