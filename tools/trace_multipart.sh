#!/bin/bash
# kernel trace of the multi-partition probe (timed steps only); idle-gap analysis of the last step
out=$GRAFT_REPO_ROOT/gpurun_out/trace_mp; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp NVT_READBACK_TIMEOUT=60 NVT_MP_ONLY_TIMED=1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/tools/multipart_probe.py ${1:-45000000} ${2:-8} 12.4 > $out/probe.log 2>&1
csv=$(find $out -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_window.py $csv ${3:-125} 60 --merge > $out/gaps.txt 2>&1
head -${4:-70} $out/gaps.txt
find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete
