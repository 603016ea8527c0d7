"""Summarise tools/sq_counters.sh: per kernel (nvt:: only), per LAUNCH averages of the SQ
counters and the derived per-key figures.

usage: python tools/sq_summarize.py <dir with insts/cycles_counter_collection.csv> <rows> out.json

`*_per_key` = wave-level instruction count * 64 / rows: the instructions of that class the
kernel's per-key path executes for ONE key of a column of `rows` rows (a kernel that sees only
part of the rows -- rp_count_kernel: the cold rows -- is still divided by the column's rows:
what it costs the column).  Cycle counters are quad-cycles summed over waves (MI355X_MICROARCH.md):
`lds_issue_stall_frac` = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES, `wait_any_frac` = SQ_WAIT_ANY /
SQ_WAVE_CYCLES (waves parked in s_waitcnt / barriers), `active_frac` = SQ_ACTIVE_INST_ANY /
SQ_WAVE_CYCLES.  `valu_issue_floor_us` / `salu_issue_floor_us`: the time one launch needs just to
ISSUE its vector (4 cycles per wave64 instruction and SIMD, 1024 SIMDs) / scalar (1 cycle, one
scalar unit per CU) instructions at 2.4 GHz."""
import collections
import csv
import glob
import json
import sys

src, rows, dst = sys.argv[1], float(sys.argv[2]), sys.argv[3]


def load(tag, work):
    """Counter sums over the FULL-SIZE dispatches of every kernel: the bench also launches the
    kernels on its small parity frames; a dispatch counts when its `work` counter is at least
    0.4 x the kernel's largest (columns of one frame differ by less than that)."""
    per = collections.defaultdict(lambda: collections.defaultdict(dict))   # kernel -> dispatch -> counters
    files = glob.glob(f"{src}/{tag}_counter_collection.csv") + glob.glob(f"{src}/*/{tag}_counter_collection.csv")
    for path in files:
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"]
            if "nvt::" not in name:
                continue
            name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            per[name][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    acc, launches = {}, {}
    for name, disp in per.items():
        top = max(d.get(work, 0.0) for d in disp.values())
        keep = [d for d in disp.values() if d.get(work, 0.0) >= 0.4 * top]
        launches[name] = len(keep)
        tot = collections.defaultdict(float)
        for d in keep:
            for k, v in d.items():
                tot[k] += v
        acc[name] = tot
    return acc, launches


insts, n_i = load("insts", "SQ_INSTS_VALU")
cycles, n_c = load("cycles", "SQ_WAVE_CYCLES")
out = {"rows_per_column": rows, "units": __doc__.split("usage:")[1].split("\n\n", 1)[1].strip(), "kernels": {}}
for name in sorted(set(insts) | set(cycles)):
    rec = {"launches": max(n_i.get(name, 0), n_c.get(name, 0))}
    li, lc = max(n_i.get(name, 1), 1), max(n_c.get(name, 1), 1)
    for k, v in insts.get(name, {}).items():
        rec[k + "_per_launch"] = round(v / li, 1)
        if k.startswith("SQ_INSTS"):
            rec[k.replace("SQ_INSTS_", "").lower() + "_per_key"] = round(v / li * 64.0 / rows, 3)
    cy = cycles.get(name, {})
    for k, v in cy.items():
        rec[k + "_per_launch"] = round(v / lc, 1)
    wc = cy.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        rec["lds_issue_stall_frac"] = round(cy.get("SQ_WAIT_INST_LDS", 0.0) / wc, 4)
        rec["issue_stall_frac"] = round(cy.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
        rec["wait_any_frac"] = round(cy.get("SQ_WAIT_ANY", 0.0) / wc, 4)
        rec["active_frac"] = round(cy.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4)
    if cy.get("SQ_ACTIVE_INST_LDS"):
        # (conflict CYCLES per quad-cycle an LDS instruction was active: a ratio of two units)
        rec["lds_bank_conflict_cycles_per_active_quad"] = round(
            cy.get("SQ_LDS_BANK_CONFLICT", 0.0) / cy["SQ_ACTIVE_INST_LDS"], 4)
    # VALU issue time of one launch if nothing else limited it: a wave64 instruction occupies its
    # 16-lane SIMD for 4 cycles; 1024 SIMDs at ~2.4 GHz
    if "SQ_INSTS_VALU_per_launch" in rec:
        rec["valu_issue_floor_us"] = round(rec["SQ_INSTS_VALU_per_launch"] * 4.0 / 1024.0 / 2400.0, 1)
    if "SQ_INSTS_SALU_per_launch" in rec:
        rec["salu_issue_floor_us"] = round(rec["SQ_INSTS_SALU_per_launch"] / 256.0 / 2400.0, 1)
    out["kernels"][name] = rec
json.dump(out, open(dst, "w"), indent=1)
top = sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES_per_launch", 0) * kv[1]["launches"])
for name, rec in top[:14]:
    print(f"{name[:56]:56s} n {rec['launches']:4d} valu/key {rec.get('valu_per_key', 0):7.2f} salu/key "
          f"{rec.get('salu_per_key', 0):6.2f} lds/key {rec.get('lds_per_key', 0):6.2f} wait_any "
          f"{rec.get('wait_any_frac', 0):.2f} issue_stall {rec.get('issue_stall_frac', 0):.2f} lds_stall "
          f"{rec.get('lds_issue_stall_frac', 0):.3f} valu_floor_us {rec.get('valu_issue_floor_us', 0):7.1f} "
          f"salu_floor_us {rec.get('salu_issue_floor_us', 0):6.1f}")
