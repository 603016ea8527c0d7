#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> gpurun_out/$1/
out=/root/repo/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python /root/repo/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $out/bench.log 2>&1
ls $out
