#!/bin/bash
# A / B / A / B of one environment switch inside one box: tools/var_ab.sh <outdir> <VAR=a> <VAR=b>
cd /root/repo; out=gpurun_out/$1; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extra > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    r=json.load(open("$out/$name.json")); fam=r["roofline"]["per_family"]
    print("$name", "$@", {k:round(r[k],2) for k in ("ms_per_step","gpu_busy_ms_per_step")}, {k: round(v["ms_per_step"],3) for k,v in fam.items()})
except Exception as e: print("$name", "FAILED", e)
PY
}
run a1 $2
run b1 $3
run a2 $2
run b2 $3
