"""Per column (separated by mt19937_folds_kernel markers): kernels of the counting passes, average
duration per pass.  usage: count_cols_parse.py <kernel_trace.csv> <repeats>"""
import collections
import csv
import sys

tr = list(csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2])
tr.sort(key=lambda r: int(r["Start_Timestamp"]))


def name(r):
    return r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("nvt::", "")[:40]


cols, cur = [], None
nmark = 0
if "--names" in sys.argv:
    print(collections.Counter(name(r) for r in tr).most_common(60))
for r in tr:
    nm = name(r)
    if "mt19937_folds" in nm:   # markers alternate: start of a column's timed passes, end of them
        nmark += 1
        if nmark % 2 == 1:
            cur = []
        else:
            cols.append(cur)
            cur = None
    elif cur is not None:
        cur.append((nm, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
tot_all = collections.Counter()
for j, c in enumerate(cols):
    agg = collections.OrderedDict()
    for nm, d, a, b in c:
        if nm.startswith("at::") or "elementwise" in nm:
            continue
        agg.setdefault(nm, [0, 0])
        agg[nm][0] += d
        agg[nm][1] += 1
    span = (max(b for _, _, _, b in c) - min(a for _, _, a, _ in c)) / reps / 1e3 if c else 0
    tot = sum(v[0] for v in agg.values()) / reps / 1e3
    short = {"rp_partition_kernel<2>": "part", "rp_count_kernel": "count", "hot_totals_kernel": "tot",
             "__amd_rocclr_fillBufferAligned": "memset", "mailbox_post_kernel": "post", "hot_sample_kernel": "sample"}
    print("C%-2d %7.1f us per pass | " % (j + 1, tot) + "  ".join(
        "%s %.1f" % (short.get(k, k.replace("_kernel", "")[:14]), v[0] / reps / 1e3) for k, v in agg.items()))
    for k, v in agg.items():
        tot_all[k] += v[0] / reps / 1e3
print("sum over columns, us:", {k: round(v, 1) for k, v in tot_all.most_common()}, "total", round(sum(tot_all.values()), 1))
