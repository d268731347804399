#!/bin/bash
# Regenerates the judged artifacts under profiles/ on the GPU box (run through gpurun from the repo root):
#   bench line, rocprofv3 --kernel-trace --stats summary of the same command, per-(kernel,grid) split,
#   PMC FETCH_SIZE / WRITE_SIZE passes over the stand-alone conv launches (separate --pmc runs).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_line.json 2> $O/bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-group > $O/bench_prof_line.json 2> $O/bench_prof.log
python scripts/trace_summary.py $O/trace > $O/kernel_by_grid.txt
cp $(find $O/trace -name "*kernel_stats.csv") $O/kernel_stats.csv
find $O/trace -name "*kernel_trace.csv" -delete
ARITH=${NND_PROFILE_ARITH:-bf16x3}
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python scripts/prof_conv.py 3 $ARITH > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python scripts/prof_conv.py 3 $ARITH > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_sq -o s -- python scripts/prof_conv.py 3 $ARITH > $O/pmc_sq.log 2>&1 || true
python scripts/pmc_summary.py $O/pmc_sq > $O/pmc_conv_sq.txt 2>&1 || true
python - <<PY
import csv, glob, collections
d = collections.OrderedDict()
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nnd::" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"].replace("void ", ""), r["Grid_Size"], r["Workgroup_Size"])
        d.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$O/pmc_conv_traffic.txt", "w") as o:
    o.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over python scripts/prof_conv.py 3 $ARITH on MI355X\n")
    o.write("# units: KB per dispatch (averaged over the dispatches of that kernel/grid); gfx950: FETCH_SIZE counts 1/2 of a 16-B/lane coalesced stream (MI355X_MICROARCH.md HBM section)\n")
    o.write("kernel | grid | wg | FETCH_SIZE_KB | WRITE_SIZE_KB\n")
    for k, v in d.items():
        m = {n: sum(x) / len(x) for n, x in v.items()}
        o.write(f"{k[0]} | {k[1]} | {k[2]} | {m.get('FETCH_SIZE', float('nan')):.1f} | {m.get('WRITE_SIZE', float('nan')):.1f}\n")
print(open("$O/pmc_conv_traffic.txt").read())
PY
find $O/pmc_fetch $O/pmc_write $O/pmc_sq -name "*.csv" -size +1M -delete
tail -1 $O/bench_line.json | python scripts/bench_summary.py
# other configurations (per-GPU work of BASELINE.json configs 3, 4, 5, both arithmetics) and the IGEV regulariser per layer
python scripts/bench_configs.py > $O/other_configs.txt 2>&1 || true
python scripts/prof_regulariser_layers.py > $O/igev_regulariser_layers.txt 2>&1 || true
python scripts/prof_split.py > $O/split_vs_fp32_68x120.txt 2>&1 || true
python scripts/prof_split.py 136 240 > $O/split_vs_fp32_136x240.txt 2>&1 || true
# CREStereo 1080x1920 cascade: per-kernel totals of 4 forwards; encoder timeline of one bf16x3 forward of the bench
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cre -o c -- python scripts/prof_cre_kernels.py bf16x3 3 > $O/cre.log 2>&1 || true
python scripts/kernel_stats_top.py $O/cre 30 > $O/cre_kernels.txt 2>&1 || true
tail -1 $O/cre.log >> $O/cre_kernels.txt
find $O/cre -name "*kernel_trace.csv" -delete
rocprofv3 --kernel-trace --output-format csv -d $O/enc -o b -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-hbm-group --no-roofline > /dev/null 2>&1 || true
python scripts/trace_encoder.py $O/enc 2 > $O/encoder_timeline_bf16x3.txt 2>&1 || true
python scripts/trace_encoder.py $O/enc -1 > $O/encoder_timeline_fp32.txt 2>&1 || true
find $O/enc -name "*kernel_trace.csv" -delete
python scripts/sweep_split_encoder.py > $O/split_encoder_shapes.txt 2>&1 || true
