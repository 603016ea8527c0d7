#!/bin/bash
# HBM traffic of the cfg4 probe (sort-path groupby): FETCH_SIZE and WRITE_SIZE in separate passes
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_cfg4; rm -rf $out; mkdir -p $out/pmc
cd /tmp && export TMPDIR=/tmp NVT_READBACK_TIMEOUT=60
P="python $GRAFT_REPO_ROOT/tools/cfg4_probe.py"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -o fetch -- $P > $out/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out -o write -- $P > $out/write.log 2>&1
cp $(find $out -name "fetch_counter_collection.csv" | head -1) $out/pmc/fetch_counter_collection.csv
cp $(find $out -name "write_counter_collection.csv" | head -1) $out/pmc/write_counter_collection.csv
cd $GRAFT_REPO_ROOT && python tools/pmc_summarize_cfg4.py $out/pmc $out/r03_cfg4_pmc_traffic.json
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete; find $out -name "*.db" -delete
