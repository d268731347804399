"""Timeline of the encoder part of one forward in a rocprofv3 --kernel-trace csv dir (stem_kernel .. corr1d_build).
    python scripts/trace_encoder.py DIR [WHICH]     WHICH: index of the forward (default -1 = the last one)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i0 = [i for i, r in enumerate(rows) if "stem_kernel" in r["Kernel_Name"]][int(sys.argv[2]) if len(sys.argv) > 2 else -1]
i1 = [i for i, r in enumerate(rows) if "corr1d_build" in r["Kernel_Name"] and i > i0][0]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    n = r["Kernel_Name"].replace("void ", "").replace("nnd::", "")[:70]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  grid {r['Grid_Size_X']:>8s} {r['Grid_Size_Y']:>3s} {r['Grid_Size_Z']:>2s} wg {r['Workgroup_Size_X']:>4s}  {n}")
