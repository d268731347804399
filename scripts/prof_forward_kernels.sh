#!/bin/bash
# Per-kernel table (launches, average, share) of whole forwards under rocprofv3 --kernel-trace:
#   bash scripts/prof_forward_kernels.sh igev   -> scripts/igev_forward_once.py (IGEV 544x960 batch 1, test backbone)
#   bash scripts/prof_forward_kernels.sh cre    -> scripts/prof_cre.py (CREStereo 1080x1920 / 20 iterations)
# Output: gpurun_out/fwd_<name>/kernels.txt
set -e
which=${1:-igev}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fwd_$which; rm -rf $O; mkdir -p $O
if [ "$which" = cre ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python scripts/prof_cre.py > $O/run.log 2>&1
else
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python scripts/igev_forward_once.py > $O/run.log 2>&1
fi
python - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(O + "/kernels.txt", "w") as out:
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
        out.write(f'{r["Name"][:100]:100s} n {int(r["Calls"]):5d} avg {float(r["AverageNs"]) / 1e3:9.1f} us  {100 * float(r["TotalDurationNs"]) / tot:5.1f} %\n')
    out.write(f"total kernel time {tot / 1e6:.2f} ms (3 forwards)\n")
PY
find $O/trace -name "*kernel_trace.csv" -delete
