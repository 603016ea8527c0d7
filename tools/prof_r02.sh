#!/bin/bash
# round-2 evidence: kernel stats + HBM traffic counters of the bench command (run on the GPU box)
out=/root/repo/gpurun_out/prof_r02; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# every kernel family alone (the mode of bench.py's third pass), so that the averages agree
# with the roofline object of the bench line
ALONE="NVT_ASYNC_FINALIZE=0 NVT_LAZY_FINALIZE=0 NVT_COUNT_STREAMS=1 NVT_FINALIZE_SERIAL=1"
env $ALONE timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $out/stats.log 2>&1
env $ALONE timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -o fetch -- python /root/repo/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extra > $out/fetch.log 2>&1
env $ALONE timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out -o write -- python /root/repo/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extra > $out/write.log 2>&1
cd /root/repo
f=$(find $out -name "stats_kernel_stats.csv" | head -1); cp $f $out/r02_kernel_stats.csv
fc=$(find $out -name "fetch_counter_collection.csv" | head -1); wc=$(find $out -name "write_counter_collection.csv" | head -1)
mkdir -p $out/pmc; cp $fc $out/pmc/fetch_counter_collection.csv; cp $wc $out/pmc/write_counter_collection.csv
python tools/pmc_summarize.py $out/pmc $out/r02_pmc_traffic.json > $out/pmc_summary.txt 2>&1
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete
tail -3 $out/stats.log; head -5 $out/pmc_summary.txt; ls $out
