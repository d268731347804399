#!/bin/bash
# Per-kernel table of one whole IGEV forward at 544x960 batch 1 (config 3, test backbone), fp16x2: rocprofv3 --kernel-trace of
# scripts/bench_configs.py-style forwards -> gpurun_out/igev_fwd/kernels.txt (kernel, launches, average, share).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/igev_fwd; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python scripts/igev_forward_once.py > $O/run.log 2>&1
python - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(O + "/kernels.txt", "w") as out:
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
        out.write(f'{r["Name"][:90]:90s} n {int(r["Calls"]):5d} avg {float(r["AverageNs"]) / 1e3:9.1f} us  {100 * float(r["TotalDurationNs"]) / tot:5.1f} %\n')
    out.write(f"total kernel time {tot / 1e6:.2f} ms\n")
PY
find $O/trace -name "*kernel_trace.csv" -delete
