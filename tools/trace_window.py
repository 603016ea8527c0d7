"""Idle gaps and a coarse family timeline of the LAST `window_ms` of a rocprofv3 --kernel-trace
CSV.  usage: trace_window.py <csv> <window_ms> [min_gap_us]"""
import collections
import csv
import sys

path, win = sys.argv[1], float(sys.argv[2])
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
tr = list(csv.DictReader(open(path)))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
end = max(int(r["End_Timestamp"]) for r in tr)
lo = end - int(win * 1e6)
tr = [r for r in tr if int(r["Start_Timestamp"]) >= lo]
t0 = int(tr[0]["Start_Timestamp"])


def name(r):
    return r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nvt::", "")[:44]


iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name(r)) for r in tr)
busy, cur_lo, cur_hi, last = 0, iv[0][0], iv[0][1], iv[0][2]
gaps = []
for a, b, nm in iv[1:]:
    if a > cur_hi:
        busy += cur_hi - cur_lo
        gaps.append((a - cur_hi, cur_hi - t0, last, nm))
        cur_lo, cur_hi = a, b
    else:
        cur_hi = max(cur_hi, b)
    if b >= cur_hi:
        last = nm
busy += cur_hi - cur_lo
wall = (max(b for _, b, _ in iv) - t0) / 1e6
print("window: wall %.2f ms, busy %.2f ms, idle %.2f ms in %d gaps" % (
    wall, busy / 1e6, wall - busy / 1e6, len(gaps)))
tot = collections.Counter()
for g, o, a, b in gaps:
    tot[(a, b)] += g
print("idle by (kernel before -> kernel after), ms:")
for (a, b), g in tot.most_common(14):
    print("  %7.3f  %s -> %s" % (g / 1e6, a, b))
print("gaps >= %.0f us:" % min_gap)
for g, o, a, b in gaps:
    if g / 1e3 >= min_gap:
        print("  @%8.2f ms  %7.1f us  %s -> %s" % (o / 1e6, g / 1e3, a, b))

agg = collections.defaultdict(lambda: [0, 0])
for r in tr:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[name(r)][0] += d
    agg[name(r)][1] += 1
print("kernels in the window (launches, total ms, avg us):")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
    print("  %-46s %5d %9.3f %9.1f" % (k, v[1], v[0] / 1e6, v[0] / v[1] / 1e3))
if "--merge" in sys.argv:
    for r in tr:
        if "merge_tile" in r["Kernel_Name"] or "merge_split" in r["Kernel_Name"]:
            print("  @%8.2f ms %8.1f us  %s grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6,
                  (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, name(r), r.get("Grid_Size", "")))
