#!/bin/bash
# like var_multi.sh, printing the encode family too
cd /root/repo; out=gpurun_out/$1; mkdir -p $out; var=$2; shift 2
for rep in 1 2; do for v in "$@"; do
  env $var=$v timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extra > $out/${v}_$rep.json 2> $out/${v}_$rep.err
  python - <<PY
import json
try:
    r=json.load(open("$out/${v}_$rep.json")); print("$var=$v", {k:round(r[k],2) for k in ("ms_per_step","gpu_busy_ms_per_step")}, r["roofline"]["per_kernel_ms_per_step"]["encode_i32"], r["roofline"]["frac"])
except Exception as e: print("$var=$v", "FAILED", e)
PY
done; done
