import json, sys
line = [l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")][-1]
r = json.loads(line)
print(f"pairs/s {r['value']:.2f}  ms/step {r['ms_per_step']:.2f}")
rf = r.get("roofline")
if rf:
    print("all convs:", {k: round(v, 3) for k, v in rf["all_convs"].items()})
    for c in rf["per_conv"]:
        print(f"  {c['conv']:26s} {c['ms']*1e3:8.1f} us  {c['gflop']:7.3f} GF  {c['tflops']:6.1f} TF")
if "cpu_baseline" in r: print(r["cpu_baseline"], r.get("parity_max_abs_vs_oracle"))
