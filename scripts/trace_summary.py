"""Summarise a rocprofv3 --kernel-trace csv of bench.py: per (kernel, grid) durations inside the loop, main-stream gaps."""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"conv_mfma_kernel<([^>]*)>", n)
    if m: return "conv<" + m.group(1).replace(" ", "") + ">"
    m = re.search(r"conv_split_kernel<([^>]*)>", n)
    if m: return "conv_split<" + m.group(1).replace(" ", "") + ">"
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:30]
# one segment per forward (it starts at its correlation build); the LAST segment that ran the bench line's arithmetic
# (conv_split kernels when any forward used them: the exact-fp32 comparison leg runs after the timed steps)
idx = [i for i, r in enumerate(rows) if "corr1d_build" in r["Kernel_Name"]]
ends = idx[1:] + [len(rows)]
segs = [(a, b) for a, b in zip(idx, ends)]
with_split = [sg for sg in segs if any("conv_split" in r["Kernel_Name"] for r in rows[sg[0]:sg[1]])]
i0, i1 = (with_split or segs)[-1]
last = max(i for i in range(i0, i1) if "mask_upsample" in rows[i]["Kernel_Name"] or "convex_upsample" in rows[i]["Kernel_Name"])
seq = [r for r in rows[i0:last + 1] if "nnd::" in r["Kernel_Name"]]
mq = [r["Queue_Id"] for r in seq if "lookup" in r["Kernel_Name"]][0]
g = collections.OrderedDict()
for r in seq:
    k = (("M " if r["Queue_Id"] == mq else "s ") + short(r["Kernel_Name"]), r["Grid_Size_X"])
    g.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot_main = 0
for k, v in g.items():
    print(f"{k[0]:34s} grid {k[1]:>7s} n {len(v):4d} avg {sum(v)/len(v):7.1f} us  sum {sum(v)/1e3:7.2f} ms")
    if k[0].startswith("M "): tot_main += sum(v)
main = [r for r in seq if r["Queue_Id"] == mq]
gaps = sum(max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) for a, b in zip(main[:-1], main[1:])) / 1e3
span = (int(main[-1]["End_Timestamp"]) - int(main[0]["Start_Timestamp"])) / 1e3
print(f"main stream: span {span/1e3:.2f} ms, kernels {tot_main/1e3:.2f} ms, gaps {gaps/1e3:.2f} ms")

# stand-alone launches of bench.py's roofline leg (nnd_profile_conv: 1 warm + 20 timed launches per layer, after the steps)
tail = [r for r in rows[last + 1:] if "conv_mfma" in r["Kernel_Name"] or "conv_split" in r["Kernel_Name"]]
g2 = collections.OrderedDict()
for r in tail:
    k = (short(r["Kernel_Name"]), r["Grid_Size_X"], r["Workgroup_Size_X"])
    g2.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("stand-alone conv launches after the timed region (bench.py roofline leg):")
for k, v in g2.items():
    print(f"  {k[0]:32s} grid {k[1]:>7s} wg {k[2]:>4s} n {len(v):4d} avg {sum(v)/len(v):7.1f} us")
