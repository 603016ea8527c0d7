#!/bin/bash
out=/root/repo/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python /root/repo/tools/probe_dense.py > $out/probe.log 2>&1
