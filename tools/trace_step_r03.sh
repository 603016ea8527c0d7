#!/bin/bash
# kernel trace of a short bench run; timeline of the last step -> gpurun_out/trace_r03/
out=$GRAFT_REPO_ROOT/gpurun_out/trace_r03; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp NVT_READBACK_TIMEOUT=60
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra > $out/bench.log 2>&1
csv=$(find $out -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_last_step.py $csv -8 --timeline > $out/timeline.txt 2>&1
head -50 $out/timeline.txt
find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete
