cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/fh
for mode in fold nofold serial; do
  case $mode in fold) unset NND_NO_FOLDED_FLOW_HEAD NND_MU_SERIAL_FOLD;; nofold) export NND_NO_FOLDED_FLOW_HEAD=1;; serial) unset NND_NO_FOLDED_FLOW_HEAD; export NND_MU_SERIAL_FOLD=1;; esac
  echo "== $mode"; python scripts/dbg/fold_prof.py 2>&1 | grep -v amdgpu.ids
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fh/prof_$mode -o p -- python scripts/dbg/fold_prof.py > gpurun_out/fh/prof_$mode.log 2>&1
  python scripts/kernel_stats_top.py gpurun_out/fh/prof_$mode 14 | grep -i "mask_up\|flow_head2\|flow_branch\|sum " | cut -c1-130
  find gpurun_out/fh/prof_$mode -name "*kernel_trace.csv" -delete
done
