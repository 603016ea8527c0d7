#!/bin/bash
# kernel trace of three back-to-back cfg2 steps with the idle gaps listed -> gpurun_out/cfg2_tl/
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/cfg2_tl
rm -rf $out; mkdir -p $out
timeout 300 python $GRAFT_REPO_ROOT/tools/cfg2_timeline.py > $out/untraced.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/prof -- \
  python $GRAFT_REPO_ROOT/tools/cfg2_timeline.py > $out/traced.log 2>&1
f=$(ls -t $out/prof/*/*kernel_trace.csv | head -1)
python $GRAFT_REPO_ROOT/tools/timeline_gaps.py $f 15 > $out/timeline.txt
tail -2 $out/untraced.log; tail -2 $out/traced.log; tail -1 $out/timeline.txt
find $out -name "*kernel_trace.csv" -delete
