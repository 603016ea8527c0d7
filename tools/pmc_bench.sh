#!/bin/bash
# HBM traffic counters for the bench kernels: FETCH_SIZE and WRITE_SIZE in separate passes
out=/root/repo/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -o fetch -- python /root/repo/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $out/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out -o write -- python /root/repo/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $out/write.log 2>&1
ls $out
