#!/bin/bash
# kernel trace of tools/segreduce_probe.py (sort-path reductions alone) -> per-kernel averages
cd /tmp && export TMPDIR=/tmp
for sk in ${SKEWS:-3}; do
  out=$GRAFT_REPO_ROOT/gpurun_out/segr_$sk
  rm -rf $out; mkdir -p $out
  SKEW=$sk timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- \
    python $GRAFT_REPO_ROOT/tools/segreduce_probe.py > $out/probe.log 2>&1
  tail -1 $out/probe.log
  f=$(ls $out/prof/*/*kernel_stats.csv | head -1)
  python $GRAFT_REPO_ROOT/tools/stats_top.py $f 7 8
done
