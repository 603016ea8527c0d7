#!/bin/bash
# step time with 1 / 2 / 3 counting streams
cd /root/repo; out=gpurun_out/$1; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extra > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    r=json.load(open("$out/$name.json")); print("$name", {k:round(r[k],2) for k in ("ms_per_step","gpu_busy_ms_per_step","profiled_pass_ms_per_step")}, r["parity_checked_rows"] if "parity_checked_rows" in r else "")
except Exception as e: print("$name", "FAILED", e)
PY
}
run s1 NVT_COUNT_STREAMS=1
run s2 NVT_COUNT_STREAMS=2
run s3 NVT_COUNT_STREAMS=3
run s1b NVT_COUNT_STREAMS=1
