"""HBM bytes per launch of the cfg4 kernels from the FETCH_SIZE / WRITE_SIZE passes of
tools/pmc_cfg4.sh (hbm = 2 * FETCH + WRITE, the gfx950 correction of pmc_summarize_r03.py; an
upper bound for kernels dominated by random 8 / 16-byte accesses).
usage: python tools/pmc_summarize_cfg4.py <dir> out.json"""
import collections, csv, json, sys

src, dst = sys.argv[1], sys.argv[2]


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "nvt::" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[name][0] += float(r["Counter_Value"]) * 1024.0
        acc[name][1] += 1
    return acc


fetch = per_kernel(f"{src}/fetch_counter_collection.csv", "FETCH_SIZE")
write = per_kernel(f"{src}/write_counter_collection.csv", "WRITE_SIZE")
# (one sort per fit: sgb_pack_kernel until round 6, os_hist_pack_kernel since; one fit more: the parity leg)
steps = max(1, sum(v[1] for k, v in fetch.items() if "sgb_pack_kernel" in k or "os_hist_pack_kernel" in k) - 1)
rows = 20_000_000
out = {"steps": steps, "rows": rows, "kernels": {}}
tot = 0.0
for name in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(name, [0.0, 0])
    w, nw = write.get(name, [0.0, 0])
    n = max(nf, nw, 1)
    out["kernels"][name] = {"launches_per_step": round(n / steps, 2),
                            "hbm_bytes_per_launch": int((2.0 * f + w) / n),
                            "hbm_bytes_per_step": int((2.0 * f + w) / steps)}
    tot += (2.0 * f + w) / steps
out["step_hbm_bytes"] = int(tot)
out["hbm_bytes_per_step"] = int(tot)
out["traffic_over_algorithmic"] = round(tot / (46.0 * rows), 2)
out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/cfg4_probe.py "
                "(20 M rows, 5 M keys: TargetEncoding kfold 5 + JoinGroupby, sort path); "
                "algorithmic bytes per step: 46 B x 20 M rows = 0.92 GB")
json.dump(out, open(dst, "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"])[:12]:
    print(f"{k[:60]:60s} {v['launches_per_step']:5.1f}/step {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch")
print("step", round(tot / 1e9, 2), "GB")
