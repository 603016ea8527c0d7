#!/bin/bash
cp nvtabular_amd/libnvt_hip.so /tmp/orig.so
for v in orig 3 4; do
  if [ $v != orig ]; then cp nvtabular_amd/libnvt_v_$v.so nvtabular_amd/libnvt_hip.so; fi
  ./tools/trace_step.sh var_$v > /dev/null
done
cp /tmp/orig.so nvtabular_amd/libnvt_hip.so
