"""Summarise the two PMC passes of tools/pmc_bench.sh into profiles/<name>.json.

usage: python tools/pmc_summarize.py gpurun_out/<dir> profiles/r01_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; per kernel we average over its
launches.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half the bytes of
a 16 B/lane coalesced stream, so hbm_bytes_corrected = 2 * FETCH + WRITE.  Calibrated on
nvt::moments_kernel (180 MB read) and nvt::fill_norm_kernel (360 MB written)."""
import collections, csv, json, sys

src, dst = sys.argv[1], sys.argv[2]


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "nvt::" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][0] += float(r["Counter_Value"]) * 1024.0
        acc[name][1] += 1
    return acc


fetch = per_kernel(f"{src}/fetch_counter_collection.csv", "FETCH_SIZE")
write = per_kernel(f"{src}/write_counter_collection.csv", "WRITE_SIZE")
kernels = {}
for name in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(name, [0.0, 0])
    w, nw = write.get(name, [0.0, 0])
    fa = f / nf if nf else 0.0
    wa = w / nw if nw else 0.0
    kernels[name] = {"launches": max(nf, nw), "fetch_size_bytes_raw": int(fa),
                     "write_size_bytes": int(wa), "hbm_bytes_corrected": int(2 * fa + wa)}
# bench.py's profiler scopes (kernel families) -> the kernels they cover; per-LAUNCH traffic of a
# scope = sum over its kernels of (bytes per kernel launch x kernel launches) / scope launches
SCOPES = {
    "encode_i32": (["nvt::encode_hot_kernel<int, long"], 1),       # one kernel launch per scope
    "fill_normalize": (["nvt::fill_norm_many_kernel<int, double>"], 1),
    "moments": (["nvt::moments_many_kernel"], 1),
}
scopes = {}
for scope, (prefixes, _) in SCOPES.items():
    tot, launches = 0.0, 0
    for name, v in kernels.items():
        if any(name.startswith(p) for p in prefixes):
            tot += v["hbm_bytes_corrected"] * v["launches"]
            launches += v["launches"]
    if launches:
        scopes[scope] = {"launches": launches, "hbm_bytes_per_launch": int(tot / launches)}
out = {
    "scopes": scopes,
    "_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 2 "
             "--warmup 2` (45 M rows); per-launch averages over all launches of the run. gfx950 "
             "correction from MI355X_MICROARCH.md section HBM: FETCH_SIZE counts half the bytes of a "
             "16 B/lane coalesced stream, so hbm_bytes_corrected = 2*FETCH + WRITE (calibrated on "
             "nvt::moments_kernel: 180 MB read; nvt::fill_norm_kernel: 360 MB written). For kernels that "
             "mix streams with random 8-byte probes (encode_hot_kernel) the doubled figure is an "
             "upper bound.",
    "kernels": kernels,
}
json.dump(out, open(dst, "w"), indent=1)
for k, v in kernels.items():
    print(f"{k[:70]:70s} {v['launches']:5d} {v['hbm_bytes_corrected']/1e6:10.1f} MB")
