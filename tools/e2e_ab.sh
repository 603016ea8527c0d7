#!/bin/bash
# end_to_end entry under a few host-side settings (one box)
cd /root/repo; out=gpurun_out/e2e_ab; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 300 python bench.py --only-extra end_to_end > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    r=json.load(open("$out/$name.json"))["extra_configs"]["end_to_end"]
    print("$name", "$@", round(r["rows_per_s"]/1e6,1), "M rows/s", [{k: x[k] for k in ("fit_s","transform_write_s","total_s")} for x in r["runs"]], r["runs"][0]["write_phases_s"])
except Exception as e: print("$name", "FAILED", e)
PY
}
for v in "$@"; do run $(echo $v | tr '= ' '__') $v; done
