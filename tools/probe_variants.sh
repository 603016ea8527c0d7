#!/bin/bash
# usage: probe_variants.sh v1 v2 ...  -- traces tools/probe_dense.py with libnvt_v_<v>.so swapped in
cp nvtabular_amd/libnvt_hip.so /tmp/orig.so
for v in orig "$@"; do
  if [ $v != orig ]; then cp nvtabular_amd/libnvt_v_$v.so nvtabular_amd/libnvt_hip.so; fi
  out=/root/repo/gpurun_out/pv_$v; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && timeout 60 rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python /root/repo/tools/probe_dense.py > $out/log.txt 2>&1)
  echo "== $v"; grep card= $out/log.txt
  python /root/repo/tools/trace_probe_summary.py $out/p_kernel_trace.csv | grep -E "lds_stage|range_merge" | awk '{print $NF}' | paste - - | awk 'NR%3==0{printf "stage %s merge %s; ", $1, $2}'; echo
done
cp /tmp/orig.so nvtabular_amd/libnvt_hip.so
