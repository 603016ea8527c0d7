#!/bin/bash
# run-to-run spread of the driver-style command inside one box: tools/var_bench.sh <outdir> [runs]
cd /root/repo; out=gpurun_out/$1; mkdir -p $out
for i in $(seq 1 ${2:-3}); do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $out/run$i.json 2> $out/run$i.err
  python - <<PY
import json
r=json.load(open("$out/run$i.json")); print("run $i", round(r["ms_per_step"],3), round(r["gpu_busy_ms_per_step"],3), r["host_timeline_ms"]["step_period"], r["device_allocs_in_timed_region"], r["roofline"]["frac"])
PY
done
