import json, sys
line = [l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")][-1]
r = json.loads(line)
print(f"pairs/s {r['value']:.2f}  ms/step {r['ms_per_step']:.2f}")
rf = r.get("roofline")
if rf:
    print("all convs:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in rf["all_convs"].items()})
    for c in rf["per_conv"]:
        print(f"  {c['conv']:30s} {c['ms']*1e3:8.1f} us  {c['gflop']:7.3f} GF  {c['tflops']:6.1f} TF (alg)" +
              (f"   in loop {c['ms_in_loop']*1e3:7.1f} us" if "ms_in_loop" in c else ""))
    for grp, rows in rf.get("hbm_group", {}).items():
        if isinstance(rows, list):
            print(grp)
            for x in rows:
                print(f"  {x['kernel']:58s} {x['us']:8.1f} us {x['algorithmic_mb']:8.1f} MB {x['gb_per_s']:7.0f} GB/s {100*x['frac_of_8tbs']:5.1f} %")
if "exact_fp32_path" in r: print("exact fp32 path:", r["exact_fp32_path"])
if "cpu_baseline" in r: print(r["cpu_baseline"], r.get("parity_max_abs_vs_oracle"))
