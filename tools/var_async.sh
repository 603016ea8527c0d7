#!/bin/bash
# unprofiled vs profiled step time under different cross-stream hand-off variants
cd /root/repo; out=gpurun_out/$1; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    r=json.load(open("$out/$name.json")); print("$name", {k:round(r[k],2) for k in ("ms_per_step","gpu_busy_ms_per_step","profiled_pass_ms_per_step")})
except Exception as e: print("$name", "FAILED", e)
PY
}
run async X=1
run async_again X=1
run sync NVT_ASYNC_FINALIZE=0
run timing NVT_EVENT_TIMING=1
run query NVT_FLUSH_QUERY=1
run serial NVT_FINALIZE_SERIAL=1
