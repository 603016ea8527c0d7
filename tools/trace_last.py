"""Per-dispatch durations from a rocprofv3 kernel_trace.csv: python trace_last.py <csv> <substr>..."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pats = sys.argv[2:]
for r in rows:
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if any(p in nm for p in pats):
        print("%-28s %9.1f us  grid %s" % (nm.split("(")[0][-28:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", "")))
