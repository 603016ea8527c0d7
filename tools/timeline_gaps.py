"""Last step of a rocprofv3 kernel-trace CSV whose steps are separated by >= 10 ms of idle device:
every launch with its offset, its duration and the idle gap in front of it.
usage: timeline_gaps.py <kernel_trace.csv> [min gap us to flag = 8]"""
import csv
import sys

tr = list(csv.DictReader(open(sys.argv[1])))
flag = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
steps, cur, last_end = [], [], None
for r in tr:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and a - last_end > 10_000_000:
        steps.append(cur)
        cur = []
    cur.append(r)
    last_end = b if last_end is None else max(last_end, b)
steps.append(cur)
step = steps[-1]
t0 = int(step[0]["Start_Timestamp"])
end = t0
busy = gaps = 0
for r in step:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, a - end)
    gaps += gap
    busy += max(0, b - max(a, end))
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nvt::", "")[:56]
    print("%9.1f us  %8.1f us  gap %7.1f %s %s" % ((a - t0) / 1e3, (b - a) / 1e3, gap / 1e3,
                                                  "<<<" if gap / 1e3 >= flag else "   ", nm))
    end = max(end, b)
print("step: wall %.3f ms, busy %.3f ms, idle %.3f ms in %d launches" % ((end - t0) / 1e6, busy / 1e6, gaps / 1e6, len(step)))
