#!/bin/bash
# rocprofv3 kernel trace of tools/probe_paths.py (families alone): per-kernel durations
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof_paths}
mkdir -p $OUT
export NVT_ASYNC_FINALIZE=0 NVT_LAZY_FINALIZE=0 NVT_COUNT_STREAMS=1 NVT_FINALIZE_SERIAL=1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python $GRAFT_REPO_ROOT/tools/probe_paths.py > $OUT/paths.log 2>&1
CSV=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/stats_top.py $CSV 4 45 > $OUT/stats.txt 2>&1
cp $CSV $OUT/kernel_stats.csv
rm -rf $OUT/prof
tail -48 $OUT/stats.txt; grep -c no_range $OUT/paths.log
