"""Kernels of the LAST forward in a rocprofv3 --kernel-trace csv directory (from the last launch of MARK to the end),
ranked by total time:  python scripts/trace_last.py DIR [MARK=stem_kernel] [N=30]"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
mark = sys.argv[2] if len(sys.argv) > 2 else "stem_kernel"
n_top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
i0 = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]][-1]
rows = rows[i0:]
g = collections.defaultdict(list)
for r in rows:
    g[r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in g.values())
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:n_top]:
    print(f"{k:72s} n {len(v):5d}  sum {sum(v) / 1e3:8.2f} ms  {100 * sum(v) / tot:5.1f} %  avg {sum(v) / len(v):8.1f} us")
print(f"kernel time {tot / 1e3:.1f} ms over a span of {span:.1f} ms, {len(rows)} launches")
