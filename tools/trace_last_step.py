"""Summarise the LAST bench step of a rocprofv3 --kernel-trace CSV (steps are delimited by the
one moments_many_kernel launch per step): wall, GPU-busy (union of kernel intervals), idle gaps,
per-kernel totals and -- with --timeline -- every launch with its start offset."""
import collections
import csv
import sys

path = sys.argv[1]
tr = list(csv.DictReader(open(path)))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(tr) if "moments_many_kernel" in r["Kernel_Name"]]
# a step = [first count kernel before the moments launch ... last kernel before the next step's first]
starts = []
for m in marks:
    i = m
    while i > 0 and ("dense" in tr[i - 1]["Kernel_Name"] or "lds_stage" in tr[i - 1]["Kernel_Name"]
                     or "part_" in tr[i - 1]["Kernel_Name"] or "range_merge" in tr[i - 1]["Kernel_Name"]
                     or "scan_" in tr[i - 1]["Kernel_Name"] or "fillBuffer" in tr[i - 1]["Kernel_Name"]):
        i -= 1
    starts.append(i)
k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].lstrip("-").isdigit() else -2
lo, hi = starts[k], starts[k + 1] if k + 1 != 0 and k + 1 < len(starts) else len(tr)
step = tr[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in step)
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
busy, cur_lo, cur_hi, gaps = 0, iv[0][0], iv[0][1], []
for a, b in iv[1:]:
    if a > cur_hi:
        busy += cur_hi - cur_lo
        gaps.append((a - cur_hi, cur_hi - t0))
        cur_lo, cur_hi = a, b
    else:
        cur_hi = max(cur_hi, b)
busy += cur_hi - cur_lo
print("step: wall %.3f ms, gpu busy (union) %.3f ms, sum of kernels %.3f ms, launches %d" % (
    (t1 - t0) / 1e6, busy / 1e6, sum(dur(r) for r in step) / 1e6, len(step)))
print("largest idle gaps (us @ offset ms):", [(round(g / 1e3, 1), round(o / 1e6, 2)) for g, o in sorted(gaps, reverse=True)[:8]])
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nvt::", "")[:60]
    agg[nm][0] += dur(r)
    agg[nm][1] += 1
for kname, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{kname:60s} {v[1]:5d} {v[0]/1e6:8.3f} ms  avg {v[0]/v[1]/1e3:8.1f} us")
if "--timeline" in sys.argv:
    for r in step:
        nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nvt::", "")[:50]
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:10.1f} us  {dur(r)/1e3:8.1f} us  q{r.get('Queue_Id','?')}  {nm}  grid {r.get('Grid_Size','')}")
