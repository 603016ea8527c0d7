#!/bin/bash
# round-5 evidence: kernel stats + HBM traffic counters of the bench command, kernel stats of the
# multi-partition entry (run on the GPU box; PMC passes separate from the stats pass)
out=$GRAFT_REPO_ROOT/gpurun_out/prof_r06; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export NVT_READBACK_TIMEOUT=60
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- $B > $out/stats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -o fetch -- $B > $out/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out -o write -- $B > $out/write.log 2>&1
NVT_MP_ONLY_TIMED=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o mp -- python $GRAFT_REPO_ROOT/tools/multipart_probe.py 45000000 8 > $out/mp.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o cfg4 -- python $GRAFT_REPO_ROOT/tools/cfg4_probe.py > $out/cfg4.log 2>&1
P4="python $GRAFT_REPO_ROOT/tools/cfg4_probe.py"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -o c4fetch -- $P4 > $out/c4fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out -o c4write -- $P4 > $out/c4write.log 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p $out/pmc4
cp $(find $out -name "c4fetch_counter_collection.csv" | head -1) $out/pmc4/fetch_counter_collection.csv
cp $(find $out -name "c4write_counter_collection.csv" | head -1) $out/pmc4/write_counter_collection.csv
python tools/pmc_summarize_cfg4.py $out/pmc4 $out/r06_cfg4_pmc_traffic.json > $out/pmc4_summary.txt 2>&1
f=$(find $out -name "stats_kernel_stats.csv" | head -1); cp $f $out/r06_kernel_stats.csv
f=$(find $out -name "mp_kernel_stats.csv" | head -1); cp $f $out/r06_multipart_kernel_stats.csv
f=$(find $out -name "cfg4_kernel_stats.csv" | head -1); cp $f $out/r06_cfg4_kernel_stats.csv
fc=$(find $out -name "fetch_counter_collection.csv" -not -path "*/pmc4/*" | head -1); wc=$(find $out -name "write_counter_collection.csv" -not -path "*/pmc4/*" | head -1)
mkdir -p $out/pmc; cp $fc $out/pmc/fetch_counter_collection.csv; cp $wc $out/pmc/write_counter_collection.csv
python tools/pmc_summarize_r06.py $out/pmc $out/r06_pmc_traffic.json > $out/pmc_summary.txt 2>&1
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete; find $out -name "*.db" -delete
grep -h '"metric"' $out/stats.log | tail -1 > $out/r06_bench_under_rocprof.json
python tools/nvt_only_stats.py $out/r06_kernel_stats.csv $out/r06_kernel_stats_nvt_only.csv
python tools/nvt_only_stats.py $out/r06_cfg4_kernel_stats.csv $out/r06_cfg4_kernel_stats_nvt_only.csv
tail -3 $out/pmc4_summary.txt; tail -2 $out/pmc_summary.txt; ls $out; python tools/stats_top.py $out/r06_kernel_stats.csv 9 14; python tools/stats_top.py $out/r06_multipart_kernel_stats.csv 9 12
