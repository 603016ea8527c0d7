#!/bin/bash
# SQ counters of the dense-count kernels in tools/probe_dense.py (PROBE_CARDS selects columns)
out=/root/repo/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $out -o p -- python /root/repo/tools/probe_dense.py > $out/log.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d $out -o q -- python /root/repo/tools/probe_dense.py >> $out/log.txt 2>&1
python - <<PY
import csv, collections
for f in ("p","q"):
    rows = list(csv.DictReader(open("$out/%s_counter_collection.csv" % f)))
    last = {}
    for r in rows:
        if "lds_stage_kernel" in r["Kernel_Name"] or "part_count_kernel" in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
            last[key] = float(r["Counter_Value"])   # last launch of the run = the timed one
    for k in sorted(last): print(k[0], k[1], "%.3g" % last[k])
PY
