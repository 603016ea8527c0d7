#!/bin/bash
out=/root/repo/gpurun_out/cold; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python /root/repo/tools/cold_fit.py > $out/log.txt 2>&1
tail -6 $out/log.txt
python - <<'PY'
import csv, collections
tr = list(csv.DictReader(open('/root/repo/gpurun_out/cold/p_kernel_trace.csv')))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
# first full-size cold step = kernels between the first and second group of 13 fill_norm launches after the small warmup (26 fill_norm in)
fn = [i for i, r in enumerate(tr) if "fill_norm_kernel" in r["Kernel_Name"]]
step = tr[fn[12] + 1: fn[25] + 1]
agg = collections.defaultdict(lambda: [0, 0, 0])
for r in step:
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[nm][0] += d; agg[nm][1] += 1; agg[nm][2] = max(agg[nm][2], d)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"{k:60s} {v[1]:4d} total {v[0]/1e6:8.2f} ms  max {v[2]/1e3:8.1f} us")
PY
