#!/bin/bash
# count_cols.sh [rows] [repeats]: per column, per kernel durations of the counting pass ALONE
out=$GRAFT_REPO_ROOT/gpurun_out/count_cols; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp NVT_READBACK_TIMEOUT=60
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/tools/count_cols_probe.py ${1:-45000000} ${2:-3} > $out/probe.log 2>&1
csv=$(find $out -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/count_cols_parse.py $csv ${2:-3} > $out/cols.txt 2>&1
cat $out/cols.txt
find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete
