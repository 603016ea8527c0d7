#!/bin/bash
# rocprofv3 kernel statistics of the cfg4 probe (TargetEncoding + JoinGroupby) -> gpurun_out/cfg4/
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/cfg4
mkdir -p $out
NVT_READBACK_TIMEOUT=60 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- \
  python $GRAFT_REPO_ROOT/tools/cfg4_probe.py > $out/probe.log 2>&1
f=$(ls $out/prof/*/*kernel_stats.csv | head -1)
cp $f $out/kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/stats_top.py $out/kernel_stats.csv 7 24
grep -A12 ms_per_step $out/probe.log
