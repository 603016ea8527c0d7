#!/bin/bash
cp nvtabular_amd/libnvt_hip.so /tmp/orig.so
for v in orig nocur; do
  if [ $v != orig ]; then cp nvtabular_amd/libnvt_v_$v.so nvtabular_amd/libnvt_hip.so; fi
  out=/root/repo/gpurun_out/var_$v; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python /root/repo/tools/dbg_time.py > $out/log.txt 2>&1)
done
cp /tmp/orig.so nvtabular_amd/libnvt_hip.so
