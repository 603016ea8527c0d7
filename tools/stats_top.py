"""Top kernels of a rocprofv3 --stats CSV: python tools/stats_top.py <kernel_stats.csv> <steps> [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    nm = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:72]
    print("%-72s calls %5s avg %9.1f us  per step %7.3f ms" % (
        nm, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / steps / 1e6))
