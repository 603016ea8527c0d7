#!/bin/bash
# end_to_end entry under a few host-side settings (one box): decode-ahead depth / reader threads
cd /root/repo; out=gpurun_out/e2e_ab; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 300 python bench.py --only-extra end_to_end > $out/$name.json 2> $out/$name.err; python - <<PY
import json
try:
    r=json.load(open("$out/$name.json"))["extra_configs"]["end_to_end"]
    print("$name", "$@", round(r["rows_per_s"]/1e6,1), "M rows/s", {k: r["runs"][0][k] for k in ("fit_s","transform_write_s","total_s")}, r["runs"][0]["write_phases_s"])
except Exception as e: print("$name", "FAILED", e)
PY
}
run base NVT_X=0
run ahead6 NVT_DECODE_AHEAD=6
run ahead6t64 NVT_DECODE_AHEAD=6 NVT_PARQUET_READ_THREADS=64
run ahead2t64 NVT_DECODE_AHEAD=2 NVT_PARQUET_READ_THREADS=64
run pyarrow NVT_PLAIN_PARQUET_READ=0
