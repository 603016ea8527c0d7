"""Summarise the PMC passes of tools/prof_r06.sh: HBM bytes per STEP of every kernel family.

usage: python tools/pmc_summarize_r06.py <dir with fetch/write_counter_collection.csv> out.json

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  gfx950 correction
(MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half the bytes of a 16 B/lane coalesced stream, so
hbm_bytes = 2 * FETCH + WRITE (calibrated in round 1 on moments: 180 MB read per column, and
fill_normalize: 360 MB written per column).  For kernels that mix streams with random 8-byte
probes (encode, range table) the doubled figure is an upper bound.  Steps in the run = launches
of fill_norm_many_kernel (one per step)."""
import collections, csv, json, sys

src, dst = sys.argv[1], sys.argv[2]
FAMILY = [
    ("count", ("lds_stage_kernel", "range_merge_kernel", "rp_partition_kernel", "rp_count_kernel",
               "hot_totals_kernel", "hot_sample_kernel", "hot_reduce_kernel", "part_", "count_kernel")),
    ("vocab_order", ("cls_scatter", "range_patch", "range_fix_prefix", "sort_small", "ord_prep",
                     "flat_build", "flat_params", "os_", "sort_", "sort2_", "enc_clear_kernel",
                     "enc_build", "enc_head_build", "class_hist_kernel")),
    ("merge", ("merge_split_kernel", "merge_tile_kernel", "merge_payload_kernel")),
    ("encode", ("encode_hot_kernel", "encode_pipe_kernel", "encode_small_kernel", "encode_kernel")),
    ("fill_normalize", ("fill_norm",)),
    ("moments", ("moments",)),
]


def family(name):
    base = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("nvt::", "")
    for fam, pats in FAMILY:
        if any(base.startswith(p) for p in pats):
            return fam
    return "other"


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "nvt::" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0] if "(anonymous" not in r["Kernel_Name"] else \
            r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        acc[name.replace("void ", "")][0] += float(r["Counter_Value"]) * 1024.0
        acc[name.replace("void ", "")][1] += 1
    return acc


fetch = per_kernel(f"{src}/fetch_counter_collection.csv", "FETCH_SIZE")
write = per_kernel(f"{src}/write_counter_collection.csv", "WRITE_SIZE")
steps_f = sum(v[1] for k, v in fetch.items() if "fill_norm_many" in k) or 1
steps_w = sum(v[1] for k, v in write.items() if "fill_norm_many" in k) or 1
kernels, fam = {}, collections.defaultdict(float)
for name in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(name, [0.0, 0])
    w, nw = write.get(name, [0.0, 0])
    per_step = 2.0 * f / steps_f + w / steps_w
    kernels[name] = {"launches_per_step": round(max(nf / steps_f, nw / steps_w), 2),
                     "fetch_raw_bytes_per_step": int(f / steps_f), "write_bytes_per_step": int(w / steps_w),
                     "hbm_bytes_per_step": int(per_step), "family": family(name)}
    fam[family(name)] += per_step
out = {
    "families": {k: int(v) for k, v in fam.items()},
    "step": int(sum(fam.values())),
    "steps_in_run": [steps_f, steps_w],
    "_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 3 "
             "--warmup 2 --no-cpu-baseline --no-extra` (45 M rows, cold step and overlap-off pass "
             "included in the average); hbm bytes = 2*FETCH + WRITE (gfx950 correction, "
             "MI355X_MICROARCH.md section HBM; an upper bound for kernels with random 8-byte probes)",
    "kernels": kernels,
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({"families": out["families"], "step": out["step"], "steps": out["steps_in_run"]}))
