#!/bin/bash
cp nvtabular_amd/libnvt_hip.so /tmp/orig.so
for v in 65_1 65_4 65_24 8_24; do
  cp nvtabular_amd/libnvt_v_$v.so nvtabular_amd/libnvt_hip.so
  echo "== agg_unroll $v"; timeout 200 python tools/probe_dense.py 2>&1 | grep -E "card= *(3|976|39043|39884406) "
done
cp /tmp/orig.so nvtabular_amd/libnvt_hip.so
