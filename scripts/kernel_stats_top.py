"""Top kernels of a rocprofv3 --kernel-trace --stats csv dir: name, calls, total / average duration, share."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
print(f"{'kernel':78s} {'calls':>6s} {'total ms':>9s} {'avg us':>8s} share")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    n = re.sub(r"\(.*", "", r["Name"].replace("void ", "").replace("nnd::", ""))[:78]
    print(f"{n:78s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:9.3f} {float(r['AverageNs']) / 1e3:8.1f} {100 * float(r['TotalDurationNs']) / tot:5.1f}%")
print(f"sum {tot / 1e6:.3f} ms")
