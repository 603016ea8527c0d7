#!/bin/bash
# kernel trace of the cfg4 step with the idle gaps listed -> gpurun_out/cfg4_tl/
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/cfg4_tl
rm -rf $out; mkdir -p $out
timeout 300 python $GRAFT_REPO_ROOT/tools/cfg4_timeline.py > $out/untraced.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/prof -- \
  python $GRAFT_REPO_ROOT/tools/cfg4_timeline.py > $out/traced.log 2>&1
f=$(ls -t $out/prof/*/*kernel_trace.csv | head -1)
python $GRAFT_REPO_ROOT/tools/timeline_gaps.py $f > $out/timeline.txt
cat $out/untraced.log | tail -3; tail -3 $out/traced.log; tail -1 $out/timeline.txt
