#!/bin/bash
# try each libnvt_hip_U*.so variant on the small-cardinality probe
cp nvtabular_amd/libnvt_hip.so /tmp/orig.so
for U in 1 2 4; do
  cp nvtabular_amd/libnvt_hip_U$U.so nvtabular_amd/libnvt_hip.so
  echo "== U=$U"; timeout 200 python tools/probe_dense.py 2>&1 | grep -E "card=  *(3|36|976|3000) "
done
cp /tmp/orig.so nvtabular_amd/libnvt_hip.so
