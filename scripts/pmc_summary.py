import csv, collections, sys, glob
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
d = collections.OrderedDict()
for f in files:
    for r in csv.DictReader(open(f)):
        if "conv_mfma" not in r["Kernel_Name"] and "conv_split" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"].replace("void nnd::", "")[:44], r["Grid_Size"], r["Workgroup_Size"])
        d.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    m = {n: sum(x) / len(x) for n, x in v.items()}
    print(k, {n: f"{x:.3g}" for n, x in m.items()})
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        print("     per-wave-cycle: " + "  ".join(f"{n[3:]}={m[n]/wc:.3f}" for n in m if n != "SQ_WAVE_CYCLES" and n.startswith("SQ_") and "INSTS" not in n and "BUSY" not in n))
    if "SQ_WAVES" in m:
        print("     per-wave insts: " + "  ".join(f"{n[9:]}={m[n]/m['SQ_WAVES']:.0f}" for n in m if "INSTS" in n))
