#!/bin/bash
# step time with 1 / 2 / 3 encode streams (3 counting streams)
cd /root/repo; out=gpurun_out/$1; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extra > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    r=json.load(open("$out/$name.json")); print("$name", {k:round(r[k],2) for k in ("ms_per_step","gpu_busy_ms_per_step","profiled_pass_ms_per_step")})
except Exception as e: print("$name", "FAILED", e)
PY
}
run e1 NVT_ENCODE_STREAMS=1
run e2 NVT_ENCODE_STREAMS=2
run e3 NVT_ENCODE_STREAMS=3
run e1b NVT_ENCODE_STREAMS=1
