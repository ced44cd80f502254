import csv, glob, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(glob.glob(d + "/*/*_kernel_stats.csv")[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    n = r["Name"].replace("uu3d::", "").replace("void ", "")
    print("%6.2f%% %8.1f us avg x%5s  %s" % (float(r["Percentage"]), float(r["AverageNs"]) / 1e3, r["Calls"], n[:100]))
print("total kernel ms per step ~ %.3f" % (tot / 1e6 / steps))
