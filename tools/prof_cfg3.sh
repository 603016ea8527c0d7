#!/bin/bash
# rocprofv3 kernel statistics of the cfg3 probe (4 columns x 45 M rows, ~36 M distinct) -> gpurun_out/cfg3/
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/cfg3
mkdir -p $out
NVT_READBACK_TIMEOUT=60 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- \
  python $GRAFT_REPO_ROOT/tools/cfg3_probe.py > $out/probe.log 2>&1
f=$(ls $out/prof/*/*kernel_stats.csv | head -1)
cp $f $out/kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/stats_top.py $out/kernel_stats.csv 1 22
tail -3 $out/probe.log
