#!/bin/bash
# old warm-up (outputs dropped at once) vs new (two output sets alive, spare blocks cached)
cd /root/repo; out=gpurun_out/$1; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extra > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    r=json.load(open("$out/$name.json")); print("$name", "$@", round(r["ms_per_step"],2), r["host_timeline_ms"]["step_period"], "allocs", r["device_allocs_in_timed_region"])
except Exception as e: print("$name", "FAILED", e)
PY
}
run old1 NVT_BENCH_COLD_ALLOCATOR=1
run new1 NVT_BENCH_COLD_ALLOCATOR=0
run old2 NVT_BENCH_COLD_ALLOCATOR=1
run new2 NVT_BENCH_COLD_ALLOCATOR=0
