"""Summarise the last bench step of a rocprofv3 --kernel-trace CSV: wall vs GPU-busy,
per-kernel totals and per-launch durations of the hot kernels."""
import collections, csv, sys

path = sys.argv[1]
tr = list(csv.DictReader(open(path)))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
fn = [i for i, r in enumerate(tr) if "fill_norm_kernel" in r["Kernel_Name"]]
start = fn[-14] + 1
step = tr[start:]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
busy = sum(dur(r) for r in step)
print("last step: wall %.2f ms, gpu busy %.2f ms, kernels %d" % ((t1 - t0) / 1e6, busy / 1e6, len(step)))
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
    agg[nm][0] += dur(r)
    agg[nm][1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"{k:70s} {v[1]:5d} {v[0]/1e6:8.3f} ms")
for pat in sys.argv[2:]:
    rows = [r for r in step if pat in r["Kernel_Name"]]
    print(pat, [round(dur(r) / 1e3) for r in rows])
