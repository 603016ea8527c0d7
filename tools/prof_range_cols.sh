#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof_rc}
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o run -- python $GRAFT_REPO_ROOT/tools/range_cols_probe.py > $OUT/log.txt 2>&1
CSV=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/trace_last.py $CSV rp_ > $OUT/trace.txt 2>&1
rm -rf $OUT/prof
grep -v "^W2026\|^E2026" $OUT/log.txt | tail -6
awk 'NR%6==5||NR%6==0' $OUT/trace.txt | tail -12
