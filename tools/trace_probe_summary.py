"""Per-launch durations (us) of the dense-count kernels in a rocprofv3 kernel trace of
tools/probe_dense.py, in launch order (3 calls per cardinality: the last is the timed one)."""
import csv, sys
tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in tr:
    nm = r["Kernel_Name"]
    if "nvt::" not in nm:
        continue
    short = nm.split("(")[0].replace("void nvt::", "")[:60]
    print(f"{short:60s} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}")
