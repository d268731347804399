"""Per (kernel, grid) durations from a rocprofv3 --kernel-trace csv directory: python scripts/trace_by_grid.py DIR"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
g = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "nnd::" not in n:
        continue
    k = (n.split("(")[0].replace("void ", "")[:48], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    g.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in g.items():
    print(f"{k[0]:50s} grid {k[1]:>9s} {k[2]:>4s} {k[3]:>3s}  n {len(v):4d}  avg {sum(v)/len(v):8.1f} us  min {min(v):8.1f}")
