"""Kernels of a rocprofv3 --kernel-trace csv directory ranked by total time (all kernels, not only nnd::):
    python scripts/trace_top.py DIR [N]"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n_top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    g[r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in g.values())
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:n_top]:
    print(f"{k:72s} n {len(v):5d}  sum {sum(v) / 1e3:8.2f} ms  {100 * sum(v) / tot:5.1f} %  avg {sum(v) / len(v):8.1f} us")
print(f"total kernel time {tot / 1e3:.1f} ms")
