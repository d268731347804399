#!/bin/bash
# PMC passes over the stand-alone loop convs (python scripts/prof_conv.py 3 <arith>), one counter set per rocprofv3 run (never with
# trace domains): HBM-side bytes (FETCH_SIZE / WRITE_SIZE), the SQ wave-cycle breakdown, and the vector-memory path (TA / TCP).
#   scripts/pmc_conv.sh <outdir> [arith]        (on the GPU box, from the repo root)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$1; ARITH=${2:-fp16x2}; mkdir -p $O
rocprofv3 -L > $O/counters_available.txt 2>&1 || true
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $O/$name -o p -- python scripts/prof_conv.py 3 $ARITH > $O/$name.log 2>&1 || echo "pass $name failed" >> $O/failed.txt; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
run sq2 SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
# (a TA_* pass aborted inside rocprofv3 on this image and hung the run: left out)
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python - <<PY
import csv, glob, collections
d = collections.OrderedDict()
for f in glob.glob("$O/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_split" not in r["Kernel_Name"] and "conv_mfma" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"].replace("void nnd::", "")[:60], r["Grid_Size"], r["Workgroup_Size"])
        d.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$O/pmc_table.txt", "w") as o:
    o.write("# rocprofv3 --pmc (one counter set per pass) over python scripts/prof_conv.py 3 $ARITH on MI355X; averages per dispatch of each (kernel, grid, workgroup)\n")
    for k, v in d.items():
        o.write(f"{k[0]} grid {k[1]} wg {k[2]}\n")
        for n in sorted(v):
            o.write(f"    {n:36s} {sum(v[n]) / len(v[n]):16.1f}\n")
print(open("$O/pmc_table.txt").read())
PY
find $O -name "*.csv" -size +1M -delete
