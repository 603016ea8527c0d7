#!/bin/bash
# SQ counters of the range-path kernels on single bench columns (tools/range_cols_probe.py)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_rc}
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY --output-format csv -d $OUT/prof -o run -- python $GRAFT_REPO_ROOT/tools/range_cols_probe.py > $OUT/log.txt 2>&1
CSV=$(find $OUT/prof -name '*counter_collection.csv' | head -1)
python - "$CSV" <<'PY' > $OUT/pmc.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if "rp_" not in nm and "lds_stage" not in nm and "encode_hot" not in nm:
        continue
    key = (nm[-30:], r["Dispatch_Id"])
    acc.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
last = {}
for (nm, d), v in acc.items():
    last.setdefault(nm, []).append(v)
for nm, lst in last.items():
    for v in lst[-15::3]:
        print(nm, {k: int(x) for k, x in v.items()})
PY
rm -rf $OUT/prof
tail -14 $OUT/pmc.txt
