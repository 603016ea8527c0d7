#!/bin/bash
# round-3 evidence: kernel stats + HBM traffic counters of the bench command (run on the GPU box)
out=$GRAFT_REPO_ROOT/gpurun_out/prof_r03; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export NVT_READBACK_TIMEOUT=60
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- $B > $out/stats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -o fetch -- $B > $out/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out -o write -- $B > $out/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o cfg4 -- python $GRAFT_REPO_ROOT/tools/cfg4_probe.py > $out/cfg4.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out -name "stats_kernel_stats.csv" | head -1); cp $f $out/r03_kernel_stats.csv
f=$(find $out -name "cfg4_kernel_stats.csv" | head -1); cp $f $out/r03_cfg4_sorted_kernel_stats.csv
fc=$(find $out -name "fetch_counter_collection.csv" | head -1); wc=$(find $out -name "write_counter_collection.csv" | head -1)
mkdir -p $out/pmc; cp $fc $out/pmc/fetch_counter_collection.csv; cp $wc $out/pmc/write_counter_collection.csv
python tools/pmc_summarize_r03.py $out/pmc $out/r03_pmc_traffic.json > $out/pmc_summary.txt 2>&1
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete; find $out -name "*.db" -delete
grep -h '"metric"' $out/stats.log | tail -1 > $out/r03_bench_under_rocprof.json
tail -2 $out/pmc_summary.txt; ls $out; python tools/stats_top.py $out/r03_kernel_stats.csv 9 14
