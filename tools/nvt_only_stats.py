"""The library's own kernels of a rocprofv3 --stats CSV (the bench's generator / property checks
launch torch kernels that dominate the raw file): python tools/nvt_only_stats.py in.csv out.csv"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "nvt::" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()) if rows else ["Name"])
    w.writeheader()
    for r in rows:
        r["Percentage"] = f"{100.0 * float(r['TotalDurationNs']) / tot:.4f}"   # of the library's time
        w.writerow(r)
