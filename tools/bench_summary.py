"""One-screen summary of a bench.py JSON line.  usage: bench_summary.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"], 3), "rows/s %.3g" % d["value"], "frac", round(d["frac_of_hbm_peak"], 4),
      "busy", d["gpu_busy_ms_per_step"], "fresh", d["fresh_fit"]["ms_per_step"], "parity",
      d.get("parity", {}).get("parity_ok"), d.get("parity", {}).get("full_frame_ok"))
print({k: (v["ms_per_step"], v["frac"]) for k, v in d["roofline"]["per_family"].items()})
if "cpu_baseline" in d:
    print("cpu", round(d["cpu_baseline"]["value"]), d["cpu_baseline"]["cores"], "eager", d.get("eager_artifacts_step_ms"),
          "tie-ref", d.get("reference_tie_break_step_ms"))
keep = ("error", "ms_per_step", "ms_per_partition", "rows_per_s", "ratio_to_single_partition_step",
        "ratio_to_scrambled_headline", "sorted_path_kept", "counting_paths", "range_overflows", "total_s",
        "ratio_to_single_partition_rows_per_s")
for k, v in d.get("extra_configs", {}).items():
    p = v.get("parity") or {}
    print(k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in keep},
          p.get("parity_ok", p.get("full_frame_ok", p.get("files_equal_in_memory_transform"))))
