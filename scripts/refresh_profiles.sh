#!/bin/bash
# Regenerates the judged artifacts under profiles/ on the GPU box (run through gpurun from the repo root; copy what it leaves in
# gpurun_out/refresh/ to profiles/r<NN>_*):
#   bench line (default flags), rocprofv3 --kernel-trace --stats summary of the same command, per-(kernel, grid) split,
#   encoder timeline, other configurations, regulariser layers, HBM-kernel table.
# PMC passes (separate rocprofv3 --pmc runs, never with trace domains): scripts/pmc_conv.sh.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_line.json 2> $O/bench.log
tail -1 $O/bench_line.json | python scripts/bench_summary.py > $O/bench_summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-group > $O/bench_line_under_rocprof.json 2> $O/bench_prof.log
python scripts/trace_summary.py $O/trace > $O/bench_kernel_by_grid.txt
python scripts/trace_encoder.py $O/trace 2 > $O/encoder_timeline.txt 2>&1 || true
cp $(find $O/trace -name "*kernel_stats.csv") $O/bench_kernel_stats.csv
find $O/trace -name "*kernel_trace.csv" -delete
echo "bench + trace done" >> $O/progress.txt
python scripts/bench_configs.py > $O/other_configs.txt 2>&1 || true
echo "configs done" >> $O/progress.txt
python scripts/prof_regulariser_layers.py fp16x2 > $O/igev_regulariser_layers_fp16x2.txt 2>&1 || true
python scripts/prof_hbm.py > $O/hbm_kernels_graph_timed.txt 2>&1 || true
python scripts/prof_split.py > $O/split_vs_fp32_68x120.txt 2>&1 || true
echo "all done" >> $O/progress.txt
