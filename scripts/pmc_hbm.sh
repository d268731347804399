#!/bin/bash
# HBM-side bytes (rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE: separate passes, never with trace domains) of the IGEV
# HBM-group kernels at their config-3 shapes (scripts/prof_hbm.py igev) -> <outdir>/pmc_hbm_table.txt: per kernel the average
# counter per dispatch and the algorithmic bytes of profiling.py beside it.
#   scripts/pmc_hbm.sh <outdir> [igev|raft]        (on the GPU box, from the repo root; default igev)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$1; G=${2:-igev}; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python scripts/prof_hbm.py $G > $O/fetch.log 2>&1 || echo "fetch pass failed" >> $O/failed.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python scripts/prof_hbm.py $G > $O/write.log 2>&1 || echo "write pass failed" >> $O/failed.txt
python - "$O" "$G" <<'PY'
import collections, csv, glob, sys
O = sys.argv[1]
d = collections.OrderedDict()
for f in glob.glob(O + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "nnd::" not in n and "igev" not in n:
            continue
        k = (n.replace("void nnd::", "").replace("nnd::", "").split("(")[0][:48], r["Grid_Size"])
        d.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(O + "/pmc_hbm_table.txt", "w") as o:
    o.write("# rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) over python scripts/prof_hbm.py " + sys.argv[2] + " on MI355X: per (kernel, grid)\n"
            "# the average per dispatch in MB (counter unit KB).  gfx950: FETCH_SIZE reports half the bytes of 16-B-per-lane streaming reads\n"
            "# (MI355X_MICROARCH.md): the column 2xFETCH applies to the kernels that read that way (pool+interleave, transposes read 4 B per lane).\n")
    o.write(f"{'kernel':50s} {'grid':>10s} {'FETCH MB':>10s} {'2xFETCH':>10s} {'WRITE MB':>10s}\n")
    for k, v in d.items():
        fe = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1) / 1024 if v["FETCH_SIZE"] else float("nan")
        wr = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1) / 1024 if v["WRITE_SIZE"] else float("nan")
        o.write(f"{k[0]:50s} {k[1]:>10s} {fe:10.1f} {2 * fe:10.1f} {wr:10.1f}\n")
print(open(O + "/pmc_hbm_table.txt").read())
PY
find $O -name "*.csv" -size +1M -delete
