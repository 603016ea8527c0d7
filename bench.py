#!/usr/bin/env python
"""bench.py -- the north-star workload on MI355X.

Workload (BASELINE.json configs[1]): synthetic Criteo-day0, N rows x (26 int32
categorical + 13 int32 continuous, Arrow validity bitmaps), resident in HBM;
one *step* = Workflow.fit + Workflow.transform of
    C1..C26 >> Categorify()   and   I1..I13 >> FillMissing() >> Normalize()
i.e. the reference's `--normalize` Criteo benchmark
(bench/examples/dask-nvtabular-criteo-benchmark.py:200-214) without parquet I/O.

Prints ONE JSON line (rank 0).  value = rows/s over all ranks (weak scaling:
every rank owns its own N-row shard; fit statistics are merged over RCCL).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# Criteo-1TB categorical cardinalities (DLRM --arch-embedding-size)
CRITEO_CARDS = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346,
                10, 2208, 11938, 155, 4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108,
                36]
N_CONT = 13
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def synth_criteo(n, device, seed=20260923, n_cat=26, n_cont=N_CONT, scramble=True):
    """Deterministic Criteo-shaped frame, generated on `device` (torch RNG).

    Categorical j: bounded power law over its Criteo cardinality (exponent cycling
    1.05..1.2), ids scrambled by an odd multiplier mod 2^31 so they are not
    frequency-ordered; null fraction cycling {0, 0.03, 0.3}.  Continuous j:
    floor(lognormal(2, 2)) as int32, null fraction 0..0.45."""
    from nvtabular_amd.device import DeviceColumn, DeviceFrame, pack_bitmap_device

    frame = DeviceFrame()
    exps = [1.05, 1.1, 1.15, 1.2]
    nulls = [0.0, 0.03, 0.3]
    for j in range(n_cat):
        g = torch.Generator(device=device).manual_seed(seed + j)
        card = float(min(CRITEO_CARDS[j % len(CRITEO_CARDS)], max(n, 3)))
        s = exps[j % 4]
        u = torch.rand(n, device=device, dtype=torch.float64, generator=g)
        x = ((card ** (1.0 - s) - 1.0) * u + 1.0) ** (1.0 / (1.0 - s))
        x = x.floor().clamp_(1, card).to(torch.int64)
        # scramble=False: ids = the frequency rank itself (dense, frequency-ordered, pre-encoded
        # ids -- the other common input; the range path's equal-width buckets assume hash-like ids)
        ids = ((x * 2654435761 + 97 * j) % (2**31)).to(torch.int32) if scramble else x.to(torch.int32)
        valid = None
        nf = nulls[j % 3]
        if nf > 0:
            m = torch.rand(n, device=device, generator=g) >= nf
            valid = pack_bitmap_device(m)
        frame[f"C{j + 1}"] = DeviceColumn(ids.contiguous(), valid)
        del u, x
    for j in range(n_cont):
        g = torch.Generator(device=device).manual_seed(seed + 100 + j)
        z = torch.randn(n, device=device, generator=g)
        v = torch.exp(2.0 + 2.0 * z).floor().clamp_(0, 2**31 - 1).to(torch.int32)
        nf = 0.45 * j / max(n_cont - 1, 1)
        valid = None
        if nf > 0:
            m = torch.rand(n, device=device, generator=g) >= nf
            valid = pack_bitmap_device(m)
        frame[f"I{j + 1}"] = DeviceColumn(v.contiguous(), valid)
        del z
    return frame


def frame_to_oracle_pandas(frame, rows):
    """First `rows` rows as the float64-with-NaN frame pandas' parquet reader gives."""
    import pandas as pd

    data = {}
    for name, col in frame.items():
        vals = col.data[:rows].cpu().numpy()
        if col.valid is not None:
            bits = np.unpackbits(col.valid[: (rows + 7) // 8].cpu().numpy(), bitorder="little")[:rows]
            vals = vals.astype("float64")
            vals[bits == 0] = np.nan
        data[name] = vals
    return pd.DataFrame(data)


def build_workflow(cat_names, cont_names, out_path):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    cats = cat_names >> ops.Categorify(out_path=out_path, defer_artifacts=True)
    conts = cont_names >> ops.FillMissing() >> ops.Normalize()
    return nvt.Workflow(cats + conts)


def _cpu_column_job(args):
    """One column of the reference's CPU path (fit + transform), run in a worker process."""
    kind, name, path, tmp = args
    import pandas as pd

    import oracle as O

    df = pd.read_parquet(path, columns=[name])
    if kind == "cat":
        paths = O.categorify_fit([df], [name], os.path.join(tmp, "cpu_" + name), tie_break="pandas")
        O.categorify_transform(df, [name], paths)
    else:
        filled = O.fill_missing(df.copy(), [name], 0)
        mom = O.custom_moments([filled], [name])
        O.normalize_transform(filled, [name], mom["mean"].to_dict(), mom["std"].to_dict())
    return name


def cpu_baseline_worker(path, tmp, procs):
    """Runs in a fresh interpreter (no GPU context): the oracle -- the pandas restatement of the
    reference's CPU path -- over the sample, one column per task on `procs` worker processes,
    the way the reference spreads per-column groupbys over dask workers.  Prints seconds."""
    import multiprocessing as mp

    import pyarrow.parquet as pq

    names = pq.ParquetFile(path).schema_arrow.names
    jobs = [("cat" if n.startswith("C") else "cont", n, path, tmp) for n in names]
    # largest cardinalities first (they dominate the makespan)
    order = {n: i for i, n in enumerate(names)}
    jobs.sort(key=lambda j: (j[0] != "cat", -CRITEO_CARDS[int(j[1][1:]) - 1] if j[0] == "cat" else order[j[1]]))
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(procs) as pool:
        list(pool.imap_unordered(_cpu_column_job, jobs))
    print(json.dumps({"seconds": time.perf_counter() - t0, "procs": procs}))


def cpu_baseline(frame, cat_names, cont_names, sample_rows, tmp):
    """The oracle (pandas restatement of the reference's CPU path) timed on this box's host
    cores over a bounded sample of the same workload, fit + transform, columns spread over
    min(#columns, #cores) worker processes; the single-process time is reported beside it."""
    import subprocess

    import oracle as O

    df = frame_to_oracle_pandas(frame, sample_rows)
    path = os.path.join(tmp, "cpu_sample.parquet")
    df.to_parquet(path, compression=None)
    procs = max(1, min(len(df.columns), os.cpu_count() or 1))
    par = None
    try:
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", path,
                              "--cpu-baseline-procs", str(procs), "--cpu-baseline-tmp", tmp],
                             capture_output=True, text=True, timeout=600)
        par = json.loads(res.stdout.strip().splitlines()[-1])
    except Exception:  # the parallel leg is best effort; the serial one below always runs
        par = None
    # serial leg: the deterministic tie rule (count desc, value asc), so that its outputs double
    # as the oracle for the GPU parity check below (the parallel leg times the literal mode)
    t0 = time.perf_counter()
    paths = O.categorify_fit([df], cat_names, os.path.join(tmp, "cpu"), tie_break="stable")
    filled = O.fill_missing(df[cont_names].copy(), cont_names, 0)
    mom = O.custom_moments([filled], cont_names)
    enc = O.categorify_transform(df, cat_names, paths)
    filled = O.fill_missing(df[cont_names].copy(), cont_names, 0)
    norm = O.normalize_transform(filled, cont_names, mom["mean"].to_dict(), mom["std"].to_dict())
    dt = time.perf_counter() - t0
    oracle_out = {"enc": enc, "norm": norm, "mom": mom}
    pandas_v = __import__("pandas").__version__
    serial = sample_rows / dt
    if par is None:
        return {"value": serial, "unit": "rows/s", "cores": 1, "kind": "port",
                "sample": f"first {sample_rows} rows of the same synthetic frame, fit+transform, "
                          f"single process pandas {pandas_v} ({os.cpu_count()} host cores visible), "
                          f"{dt:.1f} s"}, oracle_out
    return {
        "value": sample_rows / par["seconds"], "unit": "rows/s", "cores": par["procs"], "kind": "port",
        "sample": f"first {sample_rows} rows of the same synthetic frame, fit+transform, pandas "
                  f"{pandas_v}, one column per task on {par['procs']} worker processes "
                  f"({os.cpu_count()} host cores visible): {par['seconds']:.1f} s; "
                  f"single process: {dt:.1f} s = {serial:.0f} rows/s; pandas' hash groupby is "
                  f"super-linear in the distinct keys, so the rate at the full 45 M rows would be lower",
        "single_process_rows_per_s": serial,
    }, oracle_out


def parity_check(frame, cat_names, cont_names, rows, oracle_out, tmp):
    """GPU fit + transform of the first `rows` rows (the cpu_baseline sample) against the
    oracle outputs of the same rows: Categorify labels bit-exact, means / stds and normalised
    values within 1e-6 relative (BASELINE.json north_star).  Outside every timed region."""
    import nvtabular_amd as nvt

    sub = frame.slice_rows(0, rows)
    wf = build_workflow(cat_names, cont_names, os.path.join(tmp, "parity"))
    wf.fit(nvt.Dataset(sub))
    out = wf.transform(sub)
    bad = []
    for c in cat_names:
        got = out[c].data.cpu().numpy()
        if os.environ.get("NVT_BENCH_FLIP_LABEL") == c:   # (tests: a wrong label must fail the run)
            got = got.copy()
            got[len(got) // 2] += 1
        exp = oracle_out["enc"][c].to_numpy()
        if got.shape != exp.shape or not (got == exp).all():
            bad.append(c)
    mom = oracle_out["mom"]
    from nvtabular_amd.node import iter_nodes

    norm_op = [n.op for n in iter_nodes(wf.output_node) if type(n.op).__name__ == "Normalize"][0]
    worst = 0.0
    for c in cont_names:
        for got, exp in ((norm_op.means[c], float(mom["mean"][c])), (norm_op.stds[c], float(mom["std"][c]))):
            worst = max(worst, abs(got - exp) / max(abs(exp), 1e-300))
        g = out[c].data.cpu().numpy()
        e = oracle_out["norm"][c].to_numpy()
        err = float(np.max(np.abs(g - e) / np.maximum(np.abs(e), 1.0)))
        worst = max(worst, err)
    ok = not bad and worst <= 1e-6
    return {"method": "oracle (pandas restatement of the reference) on the same rows: labels bit-exact, "
                      "moments / normalised values <= 1e-6 relative",
            "parity_checked_rows": rows, "parity_ok": bool(ok), "categorify_mismatch_columns": bad,
            "normalize_max_rel_err": worst}


def std_vs_fp64(keys, y, got_std, oracle_std):
    """Both sides of the cfg4 `_std` comparison against an fp64 ground truth (numpy: per-group
    two-pass variance in float64, ddof 1) over the groups whose true std is >= 1e-2: shows that
    the 5e-3 tolerance of that column is the float32 accumulation of the pandas path, not this
    engine's arithmetic."""
    uk, inv = np.unique(keys, return_inverse=True)
    yy = y.astype("float64")
    cnt = np.bincount(inv, minlength=len(uk)).astype("float64")
    mean = np.bincount(inv, weights=yy, minlength=len(uk)) / np.maximum(cnt, 1)
    dev = yy - mean[inv]
    var = np.bincount(inv, weights=dev * dev, minlength=len(uk)) / np.maximum(cnt - 1, 1)
    true = np.sqrt(var)[inv]
    sel = (cnt[inv] >= 2) & (true >= 1e-2)
    out = {"groups": int(len(uk)), "rows_compared": int(sel.sum())}
    for name, v in (("engine", got_std), ("oracle_pandas_float32", oracle_std)):
        v = np.asarray(v, dtype="float64")
        ok = sel & ~np.isnan(v)
        out[name + "_max_rel_err"] = float(np.max(np.abs(v[ok] - true[ok]) / true[ok])) if ok.any() else None
    return out


def parity_verdicts(result):
    """Every parity leg of the line -> (all ok?, rows checked against the oracle, [failed legs]).
    A leg that is missing where one is expected, or an extra that died with an error, fails."""
    failed, rows = [], 0
    par = result.get("parity")
    if par is not None:
        rows += int(par.get("parity_checked_rows", 0))
        if par.get("parity_ok") is not True:
            failed.append("headline")
        if "full_frame_ok" in par and par["full_frame_ok"] is not True:
            failed.append("headline.full_frame")
    for key, ent in (result.get("extra_configs") or {}).items():
        if not isinstance(ent, dict) or "error" in ent:
            failed.append(key + ":error")
            continue
        p = ent.get("parity")
        if p is None:
            failed.append(key + ":no parity leg")
            continue
        ok = p.get("parity_ok", p.get("files_equal_in_memory_transform"))
        if ok is not True:
            failed.append(key)
        if "oracle" in str(p.get("method", "")):
            rows += int(p.get("parity_checked_rows", 0))
    return (not failed), rows, failed


FAMILIES = {
    # kernel family -> (scope-name prefixes, algorithmic bytes per row per column, columns)
    "count": (("dense_count_",), 4, "cat"),            # Categorify.fit: groupby-size
    "vocab_order": (("vocab_", "encode_build"), 0, "cat"),  # write_uniques + encode tables: no column bytes
    "merge": (("merge_sorted",), 0, "cat"),            # _mid_level_groupby tree merge of partial lists
    "encode": (("encode_i32", "encode_i64"), 12, "cat"),    # Categorify.transform
    "moments": (("moments",), 4, "cont"),              # Normalize.fit
    "fill_normalize": (("fill_normalize",), 12, "cont"),
}


def family_of(scope):
    for fam, (prefixes, _, _) in FAMILIES.items():
        if any(scope.startswith(p) for p in prefixes):
            return fam
    return scope


def pmc_traffic():
    """HBM bytes from the committed PMC passes (profiles/r05_pmc_traffic.json: FETCH_SIZE /
    WRITE_SIZE collected in separate `rocprofv3 --pmc` passes over this command and corrected
    per MI355X_MICROARCH.md; made by tools/prof_r05.sh + tools/pmc_summarize.py).  PMC cannot be
    sampled from inside the timed run, so these are the figures of the same command at the same
    size.  -> {"families": {family: bytes per step}, "step": bytes per step} or None."""
    path = os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        d["source"] = ("committed profile profiles/r05_pmc_traffic.json (rocprofv3 --pmc passes of "
                       "this command at this size, tools/prof_r05.sh; NOT collected in this run)")
        return d
    except Exception:
        return None


def _host_timeline(marks):
    """min / median / max over the timed steps of: step period, time inside fit (includes the
    step's host synchronisation) and time to enqueue transform."""
    if len(marks) < 2:
        return None

    def stats(v):
        v = sorted(v)
        return [round(1e3 * v[0], 3), round(1e3 * v[len(v) // 2], 3), round(1e3 * v[-1], 3)]

    period = [marks[i + 1][0] - marks[i][0] for i in range(len(marks) - 1)]
    return {"step_period": stats(period), "fit": stats([b - a for a, b, _ in marks]),
            "transform_enqueue": stats([c - b for _, b, c in marks])}


def cfg4_roofline(per_kernel_ms, rows, traffic_file="r05_cfg4_pmc_traffic.json"):
    """`roofline` block of the TargetEncoding + JoinGroupby entries: kernel families = the
    library's launch scopes (HIP events on the launch stream, K.profile_report), algorithmic
    bytes per row from SURVEY 8(d): fit = JoinGroupby.fit (4 key + 4 value) + TargetEncoding.fit
    (4 + 4 + 1 fold byte) = 17 B/row over sort + reduce + index build; transform = (4 key + 12
    float32 statistics + 4 count) + (4 key + 1 fold byte + 4) = 29 B/row over the lookup."""
    fam = {
        "fit_sort": (("groupby_sort",), 17),          # the word sort (keys + fold bits + row ids)
        "fit_reduce": (("groupby_sorted", "merge_sorted", "merge_payload"), 17),
        "fit_index": (("groupby_index", "encode_build"), 0),
        "transform_lookup": (("groupby_lookup", "te_apply"), 29),
    }
    per_family = {}
    for name, (scopes, bpr) in fam.items():
        ms = sum(v for k, v in per_kernel_ms.items() if k in scopes)
        b = bpr * rows
        per_family[name] = {"ms_per_step": round(ms, 3), "algorithmic_bytes_per_step": b,
                            "GBps": round(b / ms / 1e6, 1) if ms > 0 and b else None,
                            "frac": round(b / ms / 1e6 / HBM_PEAK_GBS, 4) if ms > 0 and b else None}
    fit_ms = sum(per_family[k]["ms_per_step"] for k in ("fit_sort", "fit_reduce", "fit_index"))
    tr_ms = per_family["transform_lookup"]["ms_per_step"]
    whole = fit_ms + tr_ms
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", traffic_file)) as f:
            t = json.load(f)
        if int(t.get("rows", 0)) == int(rows):
            traffic = t.get("hbm_bytes_per_step")
    except Exception:
        pass
    dom = max(("fit_sort", "fit_reduce", "transform_lookup"), key=lambda k: per_family[k]["ms_per_step"])
    return {
        "bound": "hbm", "kernel": dom, "achieved": per_family[dom]["GBps"], "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": per_family[dom]["frac"],
        "fit_GBps": round(17 * rows / fit_ms / 1e6, 1) if fit_ms > 0 else None,
        "transform_GBps": round(29 * rows / tr_ms / 1e6, 1) if tr_ms > 0 else None,
        "whole_step_frac_kernel_time": round(46 * rows / whole / 1e6 / HBM_PEAK_GBS, 4) if whole > 0 else None,
        "traffic": traffic,
        "traffic_source": (f"committed profile profiles/{traffic_file} (rocprofv3 --pmc passes of "
                           "tools/cfg4_probe.py at this size; NOT collected in this run)") if traffic else None,
        "per_family": per_family,
    }


def fold_generation_ms(device, rows, kfold=5, seed=42):
    """The seeded fold column of TargetEncoding (numpy's MT19937 stream regenerated on the
    device, once per (kfold, seed, device) and process: ops/target_encoding.py::_FOLD_CACHE) --
    what the FIRST fit of a process pays and the timed steps do not."""
    from nvtabular_amd.ops import target_encoding as T

    key = (int(kfold), int(seed), str(device))
    saved = T._FOLD_CACHE.pop(key, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T._fold_column(rows, kfold, seed, device)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0)
    if saved is not None and saved.numel() >= T._FOLD_CACHE[key].numel():
        T._FOLD_CACHE[key] = saved
    return round(ms, 3)


def extra_cfg4(device, tmp, rows, sample_rows, steps=5):
    """BASELINE.json configs[3] scaled to one GPU: TargetEncoding (kfold=5, fold_seed=42,
    p_smooth=20 -- the docstring example target_encoding.py:69-77) + JoinGroupby
    (count / sum / mean / std) on `rows` rows over 5 M skewed int32 ids with a float32 target,
    fit + transform, inputs resident in HBM.  Reported beside the headline number (never
    `value`): rows/s, per-kernel times, the pandas restatement on a sample, and a parity check
    of the GPU run on that sample against it (float32 outputs: 1e-5 relative)."""
    import pandas as pd

    import nvtabular_amd as nvt
    import oracle as O
    from nvtabular_amd import kernels as K
    from nvtabular_amd import ops
    from nvtabular_amd.device import DeviceColumn, DeviceFrame

    card, p = 5_000_000, 20.0
    g = torch.Generator(device=device).manual_seed(7)
    raw = (torch.rand(rows, device=device, generator=g, dtype=torch.float64) ** 3 * card).to(torch.int64)
    key = ((raw * 2654435761) % (2**31)).to(torch.int32)
    y = torch.rand(rows, device=device, generator=g, dtype=torch.float32)
    frame = DeviceFrame({"k": DeviceColumn(key), "y": DeviceColumn(y)})
    stats = ["count", "sum", "mean", "std"]

    def build(path):
        te = ["k"] >> ops.TargetEncoding("y", kfold=5, fold_seed=42, p_smooth=p, defer_artifacts=True,
                                          out_path=os.path.join(path, "te"))
        jg = ["k"] >> ops.JoinGroupby(cont_cols=["y"], stats=stats, defer_artifacts=True,
                                      out_path=os.path.join(path, "jg"))
        return nvt.Workflow(te + jg)

    wf = build(os.path.join(tmp, "cfg4"))
    ds = nvt.Dataset(frame)

    def step():
        wf.fit(ds)
        return wf.transform(frame)

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    K.profile_begin()
    out = step()
    rep = K.profile_report()
    del out
    # JoinGroupby.fit w_key + w_cont, transform w_key + 4 * 3 float32 stats + 4 count;
    # TargetEncoding.fit w_key + w_target + 1 fold byte, transform w_key + 1 + 4
    bytes_per_row = (4 + 4) + (4 + 12 + 4) + (4 + 4 + 1) + (4 + 1 + 4)
    res = {
        "workload": f"TargetEncoding(kfold=5, seed 42, p=20) + JoinGroupby(count,sum,mean,std), "
                    f"{rows} rows, 5e6 skewed int32 ids, float32 target, fit + transform",
        "rows_per_s": rows * steps / dt, "ms_per_step": 1e3 * dt / steps,
        "algorithmic_bytes_per_row": bytes_per_row,
        "algorithmic_GBps": rows * steps / dt * bytes_per_row / 1e9,
        "gpu_busy_ms": round(rep["busy_ms"], 3),
        "per_kernel_ms": {k: round(v[0], 3) for k, v in rep["kernels"].items()},
        "roofline": cfg4_roofline({k: v[0] for k, v in rep["kernels"].items()}, rows),
        # the seeded folds are a deterministic sequence cached per process: generated once,
        # OUTSIDE the timed steps (the warm-up step pays it)
        "fold_generation_ms": fold_generation_ms(device, rows),
        "fold_generation_inside_ms_per_step": False,
    }
    # ---- CPU restatement + parity on a sample ----
    m = min(sample_rows, rows)
    sub = frame.slice_rows(0, m)
    hdf = pd.DataFrame({"k": key[:m].cpu().numpy(), "y": y[:m].cpu().numpy()})
    t0 = time.perf_counter()
    cpu_dir = os.path.join(tmp, "cfg4_cpu")
    part = hdf.copy()
    te_stats, te_means = O.target_encoding_fit([part], ["k"], ["y"], os.path.join(cpu_dir, "te"),
                                               kfold=5, fold_seed=42)
    te_out = O.target_encoding_transform(hdf.copy(), ["k"], ["y"], te_stats, te_means, kfold=5,
                                         fold_seed=42, p_smooth=p)
    jg_cats = O.join_groupby_fit([hdf.copy()], [["k"]], ["y"], stats, os.path.join(cpu_dir, "jg"))
    jg_out = O.join_groupby_transform(hdf.copy(), [["k"]], jg_cats)
    cdt = time.perf_counter() - t0
    res["cpu_baseline"] = {"value": m / cdt, "unit": "rows/s", "cores": 1, "kind": "port",
                           "sample": f"first {m} rows, fit+transform, single process pandas: {cdt:.1f} s"}
    wf2 = build(os.path.join(tmp, "cfg4_par"))
    wf2.fit(nvt.Dataset(sub))
    got = wf2.transform(sub)
    worst, nan_mismatch, per_col = 0.0, {}, {}
    for name, exp in [("TE_k_y", te_out["TE_k_y"])] + [(c, jg_out[c]) for c in jg_out.columns]:
        gv = got[name].data.cpu().numpy().astype("float64")
        ev = exp.to_numpy().astype("float64")
        if name.endswith("_std"):
            # var = (sumsq - sum^2 / n) / (n - 1) cancels for groups of nearly equal values, and
            # the pandas path accumulates a float32 target in float32 (error ~1e-7 * sumsq / n
            # in var, i.e. up to ~1e-3 in a std that should be ~0): such groups compare equal
            # when both sides are below 1e-2 (or NaN from a slightly negative var)
            tiny = (np.nan_to_num(np.abs(gv), nan=0.0) < 1e-2) & (np.nan_to_num(np.abs(ev), nan=0.0) < 1e-2)
            tiny &= jg_out["k_count"].to_numpy() >= 2  # n == 1 is NaN on both sides by definition
            gv, ev = np.where(tiny, 0.0, gv), np.where(tiny, 0.0, ev)
        bad = int((np.isnan(gv) != np.isnan(ev)).sum())
        if bad:
            nan_mismatch[name] = bad
        ok = ~np.isnan(ev) & ~np.isnan(gv)
        if ok.any():
            rel = np.abs(gv[ok] - ev[ok]) / np.maximum(np.abs(ev[ok]), 1e-3)
            # float32 accumulation of the pandas path: std of a well-conditioned group agrees to
            # ~1e-4, everything else to float32 rounding
            per_col[name] = float(np.max(rel))
            worst = max(worst, per_col[name])
    # float32 outputs: 1e-5 relative; std: the pandas path accumulates the float32 target (and its
    # squares) in float32, so var = (sumsq - sum^2 / n) / (n - 1) carries a relative error of up
    # to ~1e-3 there, this engine accumulates in float64 -- 5e-3 for that column (the unit tests
    # compare float64 targets at 1e-5, tests/test_gpu_parity.py)
    tol = {c: (5e-3 if c.endswith("_std") else 1e-5) for c in per_col}
    std_cols = [c for c in jg_out.columns if c.endswith("_std")]
    res["parity"] = {"method": "oracle (pandas restatement of the reference) on the same rows",
                     "parity_checked_rows": m, "per_column_max_rel_err": per_col, "tolerance": tol,
                     "nan_mismatch": nan_mismatch,
                     "std_vs_fp64_ground_truth": {
                         c: std_vs_fp64(hdf["k"].to_numpy(), hdf["y"].to_numpy(),
                                        got[c].data.cpu().numpy(), jg_out[c].to_numpy()) for c in std_cols},
                     "parity_ok": bool(all(per_col[c] <= tol[c] for c in per_col) and not nan_mismatch)}
    return res


def extra_cfg4_multipart(device, tmp, rows_per_part=1 << 28, nparts=4, card=100_000_000, steps=2,
                         single=None):
    """BASELINE.json configs[3] at its cardinality on ONE GPU: TargetEncoding (kfold 5, seed 42,
    p_smooth 20) + JoinGroupby (count / sum / mean / std) over `nparts` partitions of
    `rows_per_part` rows, 10^8-id skewed int32 key column, float32 target, all resident.  Every
    partition's groups come from the sort path and are merged by the merge-path kernel (the fit
    stays on the sort path across partitions); transform over all partitions.  Parity: the first
    million rows of partition 0 against the oracle fitted on the same FOUR-partition prefix shape
    (4 x 250 k rows)."""
    import pandas as pd

    import nvtabular_amd as nvt
    import oracle as O
    from nvtabular_amd import kernels as K
    from nvtabular_amd import ops
    from nvtabular_amd.device import DeviceColumn, DeviceFrame

    p = 20.0
    stats = ["count", "sum", "mean", "std"]

    def make(seed, n):
        g = torch.Generator(device=device).manual_seed(seed)
        raw = (torch.rand(n, device=device, generator=g, dtype=torch.float64) ** 3 * card).to(torch.int64)
        key = ((raw * 2654435761) % (2**31)).to(torch.int32)
        y = torch.rand(n, device=device, generator=g, dtype=torch.float32)
        del raw
        return DeviceFrame({"k": DeviceColumn(key), "y": DeviceColumn(y)})

    def build(path):
        te = ["k"] >> ops.TargetEncoding("y", kfold=5, fold_seed=42, p_smooth=p, defer_artifacts=True,
                                          out_path=os.path.join(path, "te"))
        jg = ["k"] >> ops.JoinGroupby(cont_cols=["y"], stats=stats, defer_artifacts=True,
                                      out_path=os.path.join(path, "jg"))
        return nvt.Workflow(te + jg)

    import gc

    gc.collect()
    torch.cuda.empty_cache()
    frames = [make(700 + i, rows_per_part) for i in range(nparts)]
    wf = build(os.path.join(tmp, "cfg4mp"))
    ds = nvt.Dataset(frames)

    def step():
        wf.fit(ds)
        for out in wf.transform(ds).to_iter():
            del out

    # two warm-up steps: the first sizes the tables (no hints yet), the second lets the caching
    # allocator settle on the multi-GB blocks of a hinted step (a hipMalloc of several GB inside a
    # timed step is a 50-100 ms host stall: `allocator.device_allocs_in_timed_steps` says whether
    # that happened)
    step()
    step()
    torch.cuda.synchronize()
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ms1 = torch.cuda.memory_stats()
    K.profile_begin()
    step()
    rep = K.profile_report()
    from nvtabular_amd.node import iter_nodes

    jg_op = [n.op for n in iter_nodes(wf.output_node) if type(n.op).__name__ == "JoinGroupby"][0]
    st = jg_op._device_stats["k"]
    rows = rows_per_part * nparts
    res = {
        "workload": f"TargetEncoding(kfold=5, seed 42, p=20) + JoinGroupby(count,sum,mean,std), "
                    f"{nparts} partitions x {rows_per_part} rows, {card} skewed int32 ids, float32 "
                    "target, fit over all partitions + transform over all partitions, resident",
        "partitions": nparts, "rows": rows, "groups": int(st.n),
        "rows_per_s": rows / dt, "ms_per_step": 1e3 * dt, "ms_per_partition": 1e3 * dt / nparts,
        "sorted_path_kept": isinstance(st.index, K.FlatIndex),
        "single_partition_rows_per_s": single,
        "ratio_to_single_partition_rows_per_s": (rows / dt / single) if single else None,
        "gpu_busy_ms": round(rep["busy_ms"], 3),
        "rows_per_s_of_gpu_busy_time": rows / (rep["busy_ms"] / 1e3) if rep["busy_ms"] else None,
        "per_kernel_ms": {k: round(v[0], 3) for k, v in rep["kernels"].items()},
        "roofline": cfg4_roofline({k: v[0] for k, v in rep["kernels"].items()}, rows,
                                  traffic_file="r05_cfg4mp_pmc_traffic.json"),
        "fold_generation_ms": fold_generation_ms(device, rows_per_part),
        "fold_generation_inside_ms_per_step": False,
        # device allocations inside the timed steps: a step that has to hipMalloc / hipFree (the
        # caching allocator out of fitting blocks) stalls on the host, not on the GPU
        "allocator": {"reserved_GB": round(ms1["reserved_bytes.all.current"] / 1e9, 2),
                      "allocated_peak_GB": round(ms1["allocated_bytes.all.peak"] / 1e9, 2),
                      "device_allocs_in_timed_steps": ms1["num_device_alloc"] - ms0["num_device_alloc"],
                      "device_frees_in_timed_steps": ms1["num_device_free"] - ms0["num_device_free"],
                      "alloc_retries_in_timed_steps": ms1["num_alloc_retries"] - ms0["num_alloc_retries"]},
    }
    del frames, ds, wf
    torch.cuda.empty_cache()
    # parity: 4 x 250 k rows through the same multi-partition code path vs the oracle
    m = 250_000
    small = [make(900 + i, m) for i in range(4)]
    hparts = [pd.DataFrame({"k": f["k"].data.cpu().numpy(), "y": f["y"].data.cpu().numpy()}) for f in small]
    cpu_dir = os.path.join(tmp, "cfg4mp_cpu")
    te_stats, te_means = O.target_encoding_fit([h.copy() for h in hparts], ["k"], ["y"],
                                               os.path.join(cpu_dir, "te"), kfold=5, fold_seed=42)
    jg_cats = O.join_groupby_fit([h.copy() for h in hparts], [["k"]], ["y"], stats, os.path.join(cpu_dir, "jg"))
    wf2 = build(os.path.join(tmp, "cfg4mp_par"))
    wf2.fit(nvt.Dataset(small))
    worst, nan_mismatch, per_col = 0.0, {}, {}
    for h, f in zip(hparts, small):
        got = wf2.transform(f)
        te_out = O.target_encoding_transform(h.copy(), ["k"], ["y"], te_stats, te_means, kfold=5,
                                             fold_seed=42, p_smooth=p)
        jg_out = O.join_groupby_transform(h.copy(), [["k"]], jg_cats)
        for name, exp in [("TE_k_y", te_out["TE_k_y"])] + [(c, jg_out[c]) for c in jg_out.columns]:
            gv = got[name].data.cpu().numpy().astype("float64")
            ev = exp.to_numpy().astype("float64")
            if name.endswith("_std"):  # (float32 accumulation of the pandas path: see extra_cfg4)
                tiny = (np.nan_to_num(np.abs(gv), nan=0.0) < 1e-2) & (np.nan_to_num(np.abs(ev), nan=0.0) < 1e-2)
                tiny &= jg_out["k_count"].to_numpy() >= 2
                gv, ev = np.where(tiny, 0.0, gv), np.where(tiny, 0.0, ev)
            bad = int((np.isnan(gv) != np.isnan(ev)).sum())
            if bad:
                nan_mismatch[name] = nan_mismatch.get(name, 0) + bad
            ok = ~np.isnan(ev) & ~np.isnan(gv)
            if ok.any():
                rel = float(np.max(np.abs(gv[ok] - ev[ok]) / np.maximum(np.abs(ev[ok]), 1e-3)))
                per_col[name] = max(per_col.get(name, 0.0), rel)
    tol = {c: (5e-3 if c.endswith("_std") else 1e-5) for c in per_col}
    res["parity"] = {"method": "oracle (pandas restatement of the reference) on the same rows, 4 partitions",
                     "parity_checked_rows": 4 * m, "partitions": 4, "per_column_max_rel_err": per_col,
                     "tolerance": tol, "nan_mismatch": nan_mismatch,
                     "parity_ok": bool(all(per_col[c] <= tol[c] for c in per_col) and not nan_mismatch)}
    return res


LAST_TIMED = {}   # diagnostics of the last _timed_steps call


def _timed_steps(step, steps):
    """(ms per step, profile report of one more step) of a fit + transform closure."""
    from nvtabular_amd import kernels as K

    step()  # cold
    step()
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_stats()
    marks = [time.perf_counter()]
    t0 = marks[0]
    for _ in range(steps):
        out = step()
        marks.append(time.perf_counter())   # (host side: every fit holds a read-back)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    m1 = torch.cuda.memory_stats()
    # what an outlier looks like from the host: the slowest step and the allocator's traffic
    LAST_TIMED.clear()
    LAST_TIMED.update(
        max_host_step_ms=round(1e3 * max(b - a for a, b in zip(marks, marks[1:])), 3),
        device_allocs_in_timed_steps=int(m1.get("num_device_alloc", 0) - m0.get("num_device_alloc", 0)),
        device_frees_in_timed_steps=int(m1.get("num_device_free", 0) - m0.get("num_device_free", 0)),
        alloc_retries_in_timed_steps=int(m1.get("num_alloc_retries", 0) - m0.get("num_alloc_retries", 0)))
    K.profile_begin()
    out = step()
    rep = K.profile_report()
    del out
    return ms, rep


def extra_cfg3(device, tmp, rows, ncols=4, card=100_000_000, steps=3):
    """BASELINE.json configs[2]'s worst columns on one GPU: the Criteo-1TB `--high-cards`
    columns (bench/examples/dask-nvtabular-criteo-benchmark.py:360-366: 38-40 M uniques).
    `ncols` columns x `rows` rows of uniform int32 ids out of `card` (~36 M distinct per
    column at 45 M rows): nothing is hot, every row is partitioned (path 3: 64 x 256 buckets)
    and every encode probes a table in HBM.  Categorify fit + transform, HBM-resident."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops
    from nvtabular_amd.device import DeviceColumn, DeviceFrame
    from nvtabular_amd.node import iter_nodes

    frame = DeviceFrame()
    for j in range(ncols):
        g = torch.Generator(device=device).manual_seed(900 + j)
        x = torch.randint(0, card, (rows,), device=device, generator=g, dtype=torch.int64)
        frame[f"H{j}"] = DeviceColumn(((x * 2654435761 + 17 * j) % (2**31)).to(torch.int32))
        del x
    names = list(frame.columns)
    wf = nvt.Workflow(names >> ops.Categorify(out_path=os.path.join(tmp, "cfg3"), defer_artifacts=True))
    ds = nvt.Dataset(frame)

    def step():
        wf.fit(ds)
        return wf.transform(frame)

    ms, rep = _timed_steps(step, steps)
    op = [n.op for n in iter_nodes(wf.output_node) if type(n.op).__name__ == "Categorify"][0]
    keys, counts = op.fitted_vocabulary(names[0])
    bytes_per_row = ncols * (4 + 12)
    res = {
        "workload": f"{ncols} high-cardinality int32 columns x {rows} rows, uniform over {card} ids "
                    f"({int(keys[0].numel())} distinct in the first), Categorify fit + transform",
        "rows_per_s": rows / (ms / 1e3), "ms_per_step": ms,
        "algorithmic_bytes_per_row": bytes_per_row,
        "algorithmic_GBps": rows / (ms / 1e3) * bytes_per_row / 1e9,
        "counting_paths": sorted(set(op._last_paths.values())),
        "gpu_busy_ms": round(rep["busy_ms"], 3),
        "per_kernel_ms": {k: round(v[0], 3) for k, v in rep["kernels"].items()},
    }
    # parity of the whole column set through properties (the oracle cannot hold 36 M-key pandas
    # groupbys in seconds): labels are a bijection between keys and [3, 3 + distinct), counts sum
    # to the rows, the order is (count desc, key asc)
    out = wf.transform(frame)
    lab = out[names[0]].data
    n_distinct = int(keys[0].numel())
    ok = int(lab.min().item()) == 3 and int(lab.max().item()) == 2 + n_distinct
    ok = ok and bool(torch.equal(keys[0][(lab - 3)], frame[names[0]].data))
    ok = ok and int(counts.sum().item()) == rows
    c = counts
    k64 = keys[0].to(torch.int64)
    ok = ok and bool(((c[:-1] > c[1:]) | ((c[:-1] == c[1:]) & (k64[:-1] < k64[1:]))).all().item())
    res["parity"] = {"method": "properties (pandas cannot hold 36 M-key vocabularies x 4 columns in the bench's time)",
                     "property_checks": "labels <-> vocabulary bijection, counts sum to rows, "
                                        "order (count desc, key asc)", "parity_ok": bool(ok)}
    return res


def _valid_mask(col, n):
    if col.valid is None:
        return None
    idx = torch.arange(n, device=col.data.device)
    return ((col.valid[idx >> 3] >> (idx & 7).to(torch.uint8)) & 1).to(torch.bool)


def property_checks(wf, frames, outs, cat_names, cont_names):
    """Parity of the TIMED frames through size-independent properties (the oracle cannot run
    45 M-row pandas groupbys in bench time; same checks as tests/test_gpu_fullsize.py, which pin
    the labels uniquely): per categorical column the vocabulary is duplicate-free and ordered
    (count desc, key asc), its counts sum to the non-null rows, every non-null row decodes back to
    its key (vocab[label - 3] == key), nulls -> 1, nothing lands in the OOV slot, and
    bincount(labels) over ALL partitions reproduces the fit's counts; per continuous column
    mean / std equal an independent float64 torch reduction within 1e-6 relative.
    -> {"full_frame_ok": bool, "failed": [...], "rows": total rows}."""
    from nvtabular_amd.node import iter_nodes

    cat_op = [n.op for n in iter_nodes(wf.output_node) if type(n.op).__name__ == "Categorify"][0]
    norm_op = [n.op for n in iter_nodes(wf.output_node) if type(n.op).__name__ == "Normalize"]
    failed = []
    total_rows = sum(len(f) for f in frames)
    for c in cat_names:
        ks, counts = cat_op.fitted_vocabulary(c)
        keys = ks[0]
        ok = True
        if keys.numel() > 1:
            dc = counts[1:] - counts[:-1]
            ok = ok and bool((dc <= 0).all()) and bool((keys[1:][dc == 0] > keys[:-1][dc == 0]).all())
        hist = torch.zeros(keys.numel(), dtype=torch.int64, device=keys.device)
        n_valid = 0
        for frame, out in zip(frames, outs):
            col, lab = frame[c], out[c].data
            m = _valid_mask(col, len(frame))
            lv, kv = (lab, col.data) if m is None else (lab[m], col.data[m])
            n_valid += int(lv.numel())
            if m is not None:
                ok = ok and bool((lab[~m] == 1).all())
            ok = ok and int(lv.min().item()) >= 3 and int(lv.max().item()) < 3 + keys.numel()
            if not ok:
                break
            ok = ok and bool((keys[lv - 3] == kv).all())
            hist += torch.bincount(lv - 3, minlength=keys.numel())
            del lv, kv, m
        ok = ok and int(counts.sum().item()) == n_valid and bool((hist == counts).all())
        if not ok:
            failed.append(c)
        del hist
    if norm_op:
        for c in cont_names:
            cnt, sm, sq = 0.0, 0.0, 0.0
            xs = []
            for frame in frames:
                col = frame[c]
                m = _valid_mask(col, len(frame))
                x = col.data.to(torch.float64)
                xs.append(x if m is None else torch.where(m, x, torch.zeros((), dtype=torch.float64, device=x.device)))
            x = torch.cat(xs) if len(xs) > 1 else xs[0]
            mean, std = float(x.mean().item()), float(x.std(unbiased=True).item())
            del x, xs
            if not (abs(norm_op[0].means[c] - mean) <= 1e-6 * abs(mean)
                    and abs(norm_op[0].stds[c] - std) <= 1e-6 * abs(std)):
                failed.append(c)
    return {"method": "properties of the full timed frames (the oracle cannot run 45 M-row pandas groupbys "
                      "in bench time)",
            "parity_ok": not failed, "full_frame_ok": not failed, "failed": failed, "rows": total_rows,
            "checks": "vocabulary order + bijection, encode->decode round trip on every row, "
                      "bincount(labels) == fit counts, means / stds vs float64 torch within 1e-6"}


def extra_multipart(device, tmp, rows, nparts=8, steps=2, single_ms=None):
    """BASELINE.json configs[2]'s SHAPE on one GPU: the cfg2 schema as `nparts` partitions of
    `rows` rows (each its own seed: new keys keep arriving), all resident in HBM; one step =
    Workflow.fit over all partitions (per-partition groupby-size + the fan-in-8 tree merge of
    categorify.py:1054-1070,1423-1478 + moments) and Workflow.transform over all partitions
    (outputs released partition by partition).  This is the only shape Criteo-1TB can run."""
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K
    from nvtabular_amd.node import iter_nodes

    frames = [synth_criteo(rows, device, seed=31337 + 1000 * p) for p in range(nparts)]
    cat_names = [c for c in frames[0].columns if c.startswith("C")]
    cont_names = [c for c in frames[0].columns if c.startswith("I")]
    wf = build_workflow(cat_names, cont_names, os.path.join(tmp, "multipart"))
    ds = nvt.Dataset(frames)
    t_fit = []

    def step(keep=False):
        t0 = time.perf_counter()
        wf.fit(ds)
        t_fit.append(time.perf_counter() - t0)
        outs = []
        for out in wf.transform(ds).to_iter():
            if keep:
                outs.append(out)
            del out
        return outs

    step()  # cold: no hints
    step()
    torch.cuda.synchronize()
    del t_fit[:]
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    if os.environ.get("NVT_MP_ONLY_TIMED"):  # (tools/trace_multipart.sh: nothing behind the timed steps)
        return {"ms_per_step": ms, "ms_per_partition": ms / nparts}
    K.profile_begin()
    step()
    rep = K.profile_report()
    op = [n.op for n in iter_nodes(wf.output_node) if type(n.op).__name__ == "Categorify"][0]
    C, Kc = len(cat_names), len(cont_names)
    bytes_per_row = (C * 4 + Kc * 4) + (C * 12 + Kc * 12)
    fam = {}
    for scope, (tot_ms, launches, _) in rep["kernels"].items():
        f = family_of(scope)
        fam[f] = fam.get(f, 0.0) + tot_ms
    paths = {}
    for k, v in op._last_paths.items():
        paths[str(v)] = paths.get(str(v), 0) + 1
    res = {
        "workload": f"cfg2 schema as {nparts} partitions x {rows} rows (own seed each), resident; "
                    "Workflow.fit over all partitions + Workflow.transform over all partitions",
        "partitions": nparts, "rows": nparts * rows,
        "rows_per_s": nparts * rows / (ms / 1e3), "ms_per_step": ms,
        "ms_per_partition": ms / nparts,
        "fit_host_ms_per_step": round(1e3 * sum(t_fit) / max(len(t_fit), 1), 2),
        "algorithmic_GBps": nparts * rows / (ms / 1e3) * bytes_per_row / 1e9,
        "frac_of_hbm_peak": nparts * rows / (ms / 1e3) * bytes_per_row / 1e9 / HBM_PEAK_GBS,
        "single_partition_step_ms": single_ms,
        "ratio_to_single_partition_step": (ms / nparts / single_ms) if single_ms else None,
        "counting_paths": paths,
        "vocabulary_entries": int(sum(f["unique_count"] for f in op._pending.values())),
        # (column, NVT_OVF_* bits) of every range-path attempt that overflowed and was redone on the
        # sort path, over all fits of this entry
        "range_overflows": [[h, b] for h, b, _ in op._range_failures],
        "gpu_busy_ms_per_partition": round(rep["busy_ms"] / nparts, 3),
        "per_family_ms_per_partition": {k: round(v / nparts, 3) for k, v in sorted(fam.items())},
        "per_kernel_ms_per_step": {k: round(v[0], 3) for k, v in rep["kernels"].items()},
    }
    outs = step(keep=True)
    res["parity"] = property_checks(wf, frames, outs, cat_names, cont_names)
    return res


def dist_multipart(device, tmp, rows, nparts, rank, world, barrier, steps=2):
    """BASELINE.json configs[2]'s shape on N GPUs (every rank calls this): each rank holds `nparts`
    partitions of `rows` rows of the cfg2 schema (seeds differ per rank and partition), ONE exchange
    per fit whatever the number of partitions (the per-rank tree merge runs first), transform over
    the rank's partitions.  The exchange amortises over nparts partitions here -- the shape the
    north star's Criteo-1TB run has -- where the headline's N > 1 line (one partition per rank) is
    the worst case for it.  Timed like the headline: barrier + synchronize on both sides, the
    slowest rank's clock.  Compare rows_per_s with N x the N = 1 line's cfg3_multipartition."""
    import nvtabular_amd as nvt

    frames = [synth_criteo(rows, device, seed=31337 + 1000 * p + 100_000 * rank) for p in range(nparts)]
    cat_names = [c for c in frames[0].columns if c.startswith("C")]
    cont_names = [c for c in frames[0].columns if c.startswith("I")]
    wf = build_workflow(cat_names, cont_names, os.path.join(tmp, f"dist_multipart{rank}"))
    ds = nvt.Dataset(frames)

    def step():
        wf.fit(ds)
        for out in wf.transform(ds).to_iter():
            del out

    step()   # cold: no hints
    step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    if world > 1:
        import torch.distributed as td

        t = torch.tensor([ms], dtype=torch.float64, device=device if td.get_backend() == "nccl" else "cpu")
        td.all_reduce(t, op=td.ReduceOp.MAX)
        ms = float(t.item())
    del frames, ds, wf
    torch.cuda.empty_cache()
    return {
        "workload": f"cfg2 schema, {nparts} partitions x {rows} rows per rank (own seeds), resident; "
                    "Workflow.fit (one exchange per fit) + Workflow.transform over the rank's partitions",
        "partitions_per_rank": nparts, "rows_per_rank": nparts * rows, "n_gpus": world,
        "ms_per_step": ms, "rows_per_s": world * nparts * rows / (ms / 1e3),
        "ms_per_partition": ms / nparts, "scaling": "weak",
        "parity": "not checked in this entry (the N = 1 line's cfg3_multipartition and "
                  "tests/multirank_check.py cover the multi-partition and the multi-rank path)",
    }


def extra_dense_ids(device, tmp, rows, steps=5, single_ms=None):
    """The headline workload with UNSCRAMBLED ids (id = frequency rank: dense, power-law over
    [1, cardinality]): what the range path's equal-width key-range buckets cost when the keys are
    not spread like hashes (overflow -> sort path).  Same schema, same step, parity by the
    full-frame property checks."""
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K
    from nvtabular_amd.node import iter_nodes

    frame = synth_criteo(rows, device, scramble=False)
    cat_names = [c for c in frame.columns if c.startswith("C")]
    cont_names = [c for c in frame.columns if c.startswith("I")]
    wf = build_workflow(cat_names, cont_names, os.path.join(tmp, "dense_ids"))
    ds = nvt.Dataset(frame)

    def step():
        wf.fit(ds)
        return wf.transform(frame)

    ms, rep = _timed_steps(step, steps)
    op = [n.op for n in iter_nodes(wf.output_node) if type(n.op).__name__ == "Categorify"][0]
    paths = {}
    for v in op._last_paths.values():
        paths[str(v)] = paths.get(str(v), 0) + 1
    fam = {}
    for scope, (tot_ms, _, _) in rep["kernels"].items():
        fam[family_of(scope)] = fam.get(family_of(scope), 0.0) + tot_ms
    out = step()
    res = {
        "workload": f"cfg2 with dense frequency-ordered ids (no scrambling), {rows} rows, fit + transform",
        "rows_per_s": rows / (ms / 1e3), "ms_per_step": ms,
        "ratio_to_scrambled_headline": (ms / single_ms) if single_ms else None,
        "counting_paths": paths,
        "range_path_banned_columns": sorted(op._no_range),
        "range_overflows": len(op._range_failures),
        "range_overflow_bits": sorted({(h, b) for h, b, _ in op._range_failures}),
        "timed_steps": dict(LAST_TIMED),
        "per_family_ms": {k: round(v, 3) for k, v in sorted(fam.items())},
        "parity": property_checks(wf, [frame], [out], cat_names, cont_names),
    }
    return res


def reference_tie_break_step_ms(frame, cat_names, cont_names, tmp):
    """One fit + transform with Categorify(tie_break="reference"): the (key, count) lists go to the
    host and through the reference's literal two pandas sort_values calls
    (categorify.py:1300,1316), so labels equal a reference run tie for tie.  O(#uniques) host work:
    reported, never `value`."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    cats = cat_names >> ops.Categorify(out_path=os.path.join(tmp, "tie_ref"), defer_artifacts=True,
                                       tie_break="reference")
    conts = cont_names >> ops.FillMissing() >> ops.Normalize()
    wf = nvt.Workflow(cats + conts)
    ds = nvt.Dataset(frame)
    wf.fit(ds)
    out = wf.transform(frame)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wf.fit(ds)
    out = wf.transform(frame)
    torch.cuda.synchronize()
    del out
    return 1e3 * (time.perf_counter() - t0)


def extra_end_to_end(device, tmp, rows, nparts=6, reps=2):
    """What the reference's benchmark actually times
    (bench/examples/dask-nvtabular-criteo-benchmark.py:216-237): parquet files in ->
    Workflow.fit + Workflow.transform -> parquet files out, cfg2 schema, `rows` rows in `nparts`
    input files.  Input: uncompressed PLAIN parquet read by the hand-written reader (page walk +
    definition levels in host C, nvt_pq_decode_chunk, a pool thread per column; packed values over
    PCIe from pinned staging on a side stream; columns with nulls expanded on the device); output:
    the hand-written PLAIN writer (parquet_plain.py), one file per input partition.  Files live
    under the bench's temp directory (page cache)."""
    import shutil

    import pyarrow.parquet as pq

    import nvtabular_amd as nvt
    from nvtabular_amd import io as nio

    frame = synth_criteo(rows, device)
    cat_names = [c for c in frame.columns if c.startswith("C")]
    cont_names = [c for c in frame.columns if c.startswith("I")]
    in_dir, out_dir = os.path.join(tmp, "e2e_in"), os.path.join(tmp, "e2e_out")
    cuts = [(rows * i // nparts) // 8 * 8 for i in range(nparts)] + [rows]
    parts = [frame.slice_rows(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    nvt.Dataset(parts).to_parquet(in_dir)
    in_bytes = sum(os.path.getsize(os.path.join(in_dir, f)) for f in os.listdir(in_dir))
    wf = build_workflow(cat_names, cont_names, os.path.join(tmp, "e2e_wf"))
    from nvtabular_amd import parquet_plain as _pp

    def timed(src, n):
        out = []
        for _ in range(n):
            shutil.rmtree(out_dir, ignore_errors=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ds = nvt.Dataset(src, engine="parquet", row_groups_per_part=2)
            wf.fit(ds)
            t1 = time.perf_counter()
            wf.transform(ds).to_parquet(out_dir)
            t2 = time.perf_counter()
            out.append({"fit_s": t1 - t0, "transform_write_s": t2 - t1, "total_s": t2 - t0,
                        "write_phases_s": {k: round(v, 3) for k, v in nio.LAST_TIMING.items()}})
        return out

    chunks0 = dict(_pp.READER_CHUNKS)
    runs = timed(in_dir, reps + 1)
    chunks1 = dict(_pp.READER_CHUNKS)
    out_bytes = sum(os.path.getsize(os.path.join(out_dir, f)) for f in os.listdir(out_dir)
                    if f.endswith(".parquet"))
    best = min(runs[1:], key=lambda r: r["total_s"])
    # parity: the files' contents == the in-memory transform of the same rows (first input file)
    first = sorted(f for f in os.listdir(out_dir) if f.endswith(".parquet"))[0]
    got = pq.read_table(os.path.join(out_dir, first)).to_pandas()
    exp = wf.transform(parts[0]).to_pandas()
    ok = list(got.columns) == list(exp.columns) and len(got) == len(exp)
    for c in (exp.columns if ok else []):
        a, b = got[c].to_numpy(), exp[c].to_numpy()
        ok = ok and bool(((a == b) | ((a != a) & (b != b))).all())
    # the same rows as pandas / pyarrow / cuDF / the reference write them by default: snappy pages,
    # dictionary-encoded where the dictionary fits (pyarrow's writer), read by the same reader
    # (snappy blocks + RLE_DICTIONARY indices decoded in the per-column host task)
    default_files = None
    try:
        in2 = os.path.join(tmp, "e2e_in_default")
        nvt.Dataset(parts).to_parquet(in2, compression="snappy")
        in2_bytes = sum(os.path.getsize(os.path.join(in2, f)) for f in os.listdir(in2))
        c0 = dict(_pp.READER_CHUNKS)
        r2 = timed(in2, 2)
        c1 = dict(_pp.READER_CHUNKS)
        b2 = min(r2[1:], key=lambda r: r["total_s"])
        default_files = {
            "input": "pyarrow writer defaults: snappy, dictionary pages (RLE_DICTIONARY) with PLAIN fall-back",
            "rows_per_s": rows / b2["total_s"], "total_s": round(b2["total_s"], 3),
            "fit_s": round(b2["fit_s"], 3), "transform_write_s": round(b2["transform_write_s"], 3),
            "input_bytes": in2_bytes,
            "reader_column_chunks": {k: c1[k] - c0[k] for k in c1},
        }
        shutil.rmtree(in2, ignore_errors=True)
    except Exception as e:  # noqa: BLE001
        default_files = {"error": repr(e)}
    res = {
        "workload": f"cfg2 schema, {rows} rows in {nparts} uncompressed parquet files -> Workflow.fit + "
                    "transform -> parquet files (PLAIN, uncompressed), files in the page cache",
        "reader_column_chunks": {k: chunks1[k] - chunks0[k] for k in chunks1},
        "default_files": default_files,
        "rows_per_s": rows / best["total_s"], "total_s": round(best["total_s"], 3),
        "fit_rows_per_s": rows / best["fit_s"], "transform_write_rows_per_s": rows / best["transform_write_s"],
        "input_bytes": in_bytes, "output_bytes": out_bytes,
        "GBps_in_plus_out": (2 * in_bytes + out_bytes) / best["total_s"] / 1e9,
        "first_run_s": round(runs[0]["total_s"], 3),
        "runs": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in runs[1:]],
        "host_cores": os.cpu_count(),
        "parity": {"method": "the output files against the in-memory transform of the same rows (which the "
                             "headline's oracle leg covers)",
                   "files_equal_in_memory_transform": bool(ok), "checked_rows": int(len(exp))},
    }
    shutil.rmtree(in_dir, ignore_errors=True)
    shutil.rmtree(out_dir, ignore_errors=True)
    return res


def extra_cfg5(device, tmp, rows=10_000_000, steps=3):
    """BASELINE.json configs[4]: multi-hot list<int32> column (0..8 leaves per row, Zipf over
    1 M ids) + a scalar id column: Categorify on both and HashBucket on the list column
    (categorify.py:1696,1803; hash_bucket.py:93-96 act on the leaves and keep the offsets).
    The list path reuses the scalar kernels on the leaves; parity against the oracle on a
    sample."""
    import pandas as pd

    import nvtabular_amd as nvt
    import oracle as O
    from nvtabular_amd import ops
    from nvtabular_amd.device import DeviceColumn, DeviceFrame

    g = torch.Generator(device=device).manual_seed(55)
    lens = torch.randint(0, 9, (rows,), device=device, generator=g, dtype=torch.int64)
    offsets = torch.zeros(rows + 1, dtype=torch.int64, device=device)
    offsets[1:] = torch.cumsum(lens, 0)
    nleaf = int(offsets[-1].item())
    u = torch.rand(nleaf, device=device, dtype=torch.float64, generator=g)
    s, card = 1.15, 1.0e6
    x = (((card ** (1.0 - s) - 1.0) * u + 1.0) ** (1.0 / (1.0 - s))).floor().clamp_(1, card).to(torch.int64)
    leaves = ((x * 2654435761) % (2**31)).to(torch.int32)
    item = torch.randint(0, 50_000, (rows,), device=device, generator=g, dtype=torch.int32)
    frame = DeviceFrame({"tags": DeviceColumn(leaves, None, offsets), "item": DeviceColumn(item)})

    def build(path):
        cats = ["tags", "item"] >> ops.Categorify(out_path=path, defer_artifacts=True)
        hb = ["tags"] >> ops.HashBucket(1000) >> ops.Rename(postfix="_hb")
        return nvt.Workflow(cats + hb)

    wf = build(os.path.join(tmp, "cfg5"))
    ds = nvt.Dataset(frame)

    def step():
        wf.fit(ds)
        return wf.transform(frame)

    ms, rep = _timed_steps(step, steps)
    bytes_per_row_leaf = 4 + 12 + 4 + 4  # categorify fit + transform, hash bucket in + out
    res = {
        "workload": f"{rows} rows, list<int32> column with {nleaf} leaves (Zipf over 1e6 ids) + "
                    "scalar id column: Categorify (both) + HashBucket(1000) (list), fit + transform",
        "rows_per_s": rows / (ms / 1e3), "leaves_per_s": nleaf / (ms / 1e3), "ms_per_step": ms,
        "algorithmic_GBps": (nleaf * bytes_per_row_leaf + rows * 16) / (ms / 1e3) / 1e9,
        "gpu_busy_ms": round(rep["busy_ms"], 3),
        "per_kernel_ms": {k: round(v[0], 3) for k, v in rep["kernels"].items()},
    }
    # parity on the first 200 k rows against the oracle
    m = 200_000
    hoff = offsets[: m + 1].cpu().numpy()
    hleaf = leaves[: int(hoff[-1])].cpu().numpy()
    hdf = pd.DataFrame({"tags": [hleaf[a:b] for a, b in zip(hoff[:-1], hoff[1:])],
                        "item": item[:m].cpu().numpy()})
    sub = DeviceFrame({"tags": DeviceColumn(leaves[: int(hoff[-1])].contiguous(), None,
                                            offsets[: m + 1].contiguous()),
                       "item": DeviceColumn(item[:m].contiguous())})
    wf2 = build(os.path.join(tmp, "cfg5_par"))
    wf2.fit(nvt.Dataset(sub))
    got = wf2.transform(sub)
    paths = O.categorify_fit([hdf], ["tags", "item"], os.path.join(tmp, "cfg5_cpu"), tie_break="stable")
    exp = O.categorify_transform(hdf, ["tags", "item"], paths)
    exp_leaves = np.concatenate([np.asarray(r) for r in exp["tags"]]) if m else np.empty(0)
    ok = bool((got["tags"].data.cpu().numpy() == exp_leaves).all())
    ok = ok and bool((got["item"].data.cpu().numpy() == exp["item"].to_numpy()).all())
    hb_exp = np.concatenate([np.asarray(r) for r in
                             O.hash_bucket_op(hdf[["tags"]].copy(), 1000, cols=["tags"])["tags"]])
    ok = ok and bool((got["tags_hb"].data.cpu().numpy() == hb_exp).all())
    res["parity"] = {"method": "oracle (pandas restatement of the reference) on the first rows",
                     "parity_checked_rows": m, "parity_ok": ok}
    return res


def eager_artifacts_step_ms(frame, cat_names, cont_names, tmp, steps=2):
    """The drop-in default: Categorify(defer_artifacts=False) writes unique.*/meta.*.parquet
    inside fit like the reference (categorify.py:1326-1334).  ms per fit + transform step."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    cats = cat_names >> ops.Categorify(out_path=os.path.join(tmp, "eager"))
    conts = cont_names >> ops.FillMissing() >> ops.Normalize()
    wf = nvt.Workflow(cats + conts)
    ds = nvt.Dataset(frame)
    wf.fit(ds)
    out = wf.transform(frame)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wf.fit(ds)
        out = wf.transform(frame)
    torch.cuda.synchronize()
    del out
    return 1e3 * (time.perf_counter() - t0) / steps


# the collective FORMS dist.py relies on, by name (tests assert on this set, not on a count)
SELFCHECK_COLLECTIVES = ("all_to_all_single(uneven)", "all_gather_v", "all_gather_v(int32)",
                         "all_gather(equal shapes)", "all_reduce(MAX)", "all_reduce(SUM)")


def collective_selfcheck(device, backend):
    """Before anything is timed on more than one rank: the collective FORMS dist.py relies on
    (all_to_all_single with uneven splits, the all-gather(v) written as an all-to-all in which
    every rank sends its whole tensor to every peer, all_gather of equal shapes, MAX / SUM
    all-reduce) run once on tiny tensors over the measured backend (nccl = RCCL) AND over a gloo
    group of the same processes, and must agree element for element.  A wiring problem (IPC mode,
    device binding, split-size semantics) then shows up as `collective_selfcheck` failing with
    its reason, not as a hang or a wrong vocabulary minutes into the fit."""
    import torch.distributed as td

    G, r = td.get_world_size(), td.get_rank()
    ref = td.new_group(backend="gloo")
    report = {"backend": backend, "world": G, "ok": False, "checks": []}

    def both(fn):
        # (gloo rehearsal on one GPU, NVT_BENCH_BACKEND=gloo: both legs are host legs)
        dev_out = fn(device if backend == "nccl" else torch.device("cpu"), None)
        cpu_out = fn(torch.device("cpu"), ref)
        return dev_out.cpu(), cpu_out

    def a2a_uneven(dev, group):
        send_counts = [(r + 1) * (p + 2) for p in range(G)]           # rows this rank sends to p
        recv_counts = [(p + 1) * (r + 2) for p in range(G)]
        send = torch.cat([torch.full((c,), 1000 * r + p, dtype=torch.int64) for p, c in enumerate(send_counts)]).to(dev)
        out = torch.empty(sum(recv_counts), dtype=torch.int64, device=dev)
        if dev.type == "cpu":   # gloo has no all_to_all_single on every build: pairwise reference
            reqs = [td.isend(send[sum(send_counts[:p]): sum(send_counts[:p + 1])].contiguous(), p, group=group)
                    for p in range(G) if p != r]
            for p in range(G):
                lo = sum(recv_counts[:p])
                if p == r:
                    out[lo: lo + recv_counts[p]] = send[sum(send_counts[:r]): sum(send_counts[:r + 1])]
                else:
                    buf = torch.empty(recv_counts[p], dtype=torch.int64)
                    td.recv(buf, p, group=group)
                    out[lo: lo + recv_counts[p]] = buf
            for q in reqs:
                q.wait()
        else:
            td.all_to_all_single(out, send, output_split_sizes=recv_counts, input_split_sizes=send_counts)
        return out

    def gather_v(dev, group, dtype=torch.int64):
        mine = torch.arange(3 + 2 * r, dtype=dtype, device=dev) + 100 * r
        sizes = [3 + 2 * p for p in range(G)]
        if dev.type == "cpu":
            pad = torch.zeros(max(sizes), dtype=dtype)
            pad[: mine.numel()] = mine
            bufs = [torch.empty_like(pad) for _ in range(G)]
            td.all_gather(bufs, pad, group=group)
            return torch.cat([b[:s] for b, s in zip(bufs, sizes)])
        out = torch.empty(sum(sizes), dtype=dtype, device=dev)
        td.all_to_all_single(out, mine.repeat(G), output_split_sizes=sizes, input_split_sizes=[mine.numel()] * G)
        return out

    def gather_v_i32(dev, group):   # (the labels of the distributed ordering travel as int32)
        return gather_v(dev, group, torch.int32)

    def gather_equal(dev, group):   # (class histograms + lengths + scalars: one int64 matrix per rank)
        mine = (torch.arange(15, dtype=torch.int64).reshape(3, 5) * (r + 1)).to(dev)
        bufs = [torch.empty_like(mine) for _ in range(G)]
        td.all_gather(bufs, mine, group=group)
        return torch.stack(bufs)

    def reduce_max(dev, group):
        t = torch.tensor([r, -r, 7], dtype=torch.int64, device=dev)
        td.all_reduce(t, op=td.ReduceOp.MAX, group=group)
        return t

    def reduce_sum(dev, group):
        t = torch.tensor([1.5 * (r + 1), 2.0], dtype=torch.float64, device=dev)
        td.all_reduce(t, op=td.ReduceOp.SUM, group=group)
        return t

    try:
        forms = (a2a_uneven, gather_v, gather_v_i32, gather_equal, reduce_max, reduce_sum)
        assert len(forms) == len(SELFCHECK_COLLECTIVES)
        for name, fn in zip(SELFCHECK_COLLECTIVES, forms):
            a, b = both(fn)
            same = a.shape == b.shape and bool(torch.equal(a, b))
            report["checks"].append({"collective": name, "equal_to_gloo": same})
        report["ok"] = all(c["equal_to_gloo"] for c in report["checks"])
    except Exception as e:
        report["error"] = repr(e)
    flag = torch.tensor([1 if report["ok"] else 0], dtype=torch.int64)
    td.all_reduce(flag, op=td.ReduceOp.MIN, group=ref)
    report["ok_on_every_rank"] = bool(int(flag.item()))
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=45_000_000, help="rows per GPU")
    ap.add_argument("--cpu-sample", type=int, default=5_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[3] (TE + JoinGroupby) entry")
    ap.add_argument("--cfg4-rows", type=int, default=20_000_000)
    ap.add_argument("--multipart", type=int, default=8, help="partitions of the cfg3_multipartition entry")
    ap.add_argument("--cfg4-parts", type=int, default=4)
    ap.add_argument("--cfg4-part-rows", type=int, default=1 << 28)
    ap.add_argument("--only-extra", default=None, help="run only this extra_configs entry (diagnostic)")
    ap.add_argument("--cpu-baseline-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-procs", type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-tmp", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args.cpu_baseline_worker, args.cpu_baseline_tmp or tempfile.mkdtemp(),
                            args.cpu_baseline_procs)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched bare: start one rank per GPU ourselves (the reference's benchmark also
        # brings up its own workers, bench/examples/dask-nvtabular-criteo-benchmark.py:176-194)
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port",
               str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # one rank per GPU; NVT_BENCH_SHARE_GPU=1 folds ranks onto the visible devices (debug only)
    dev_index = local_rank % torch.cuda.device_count() if os.environ.get("NVT_BENCH_SHARE_GPU") \
        else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = None
    if world > 1:
        import torch.distributed as td

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # NVT_BENCH_BACKEND=gloo (+ NVT_BENCH_SHARE_GPU=1): end-to-end rehearsal of the multi-rank
        # code on a single GPU, collectives staged through the host -- a correctness tool, its
        # timings mean nothing.  The measured configuration is nccl (= RCCL), one rank per GPU.
        backend = os.environ.get("NVT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            td.init_process_group(backend, rank=rank, world_size=world)
        assert td.get_world_size() == args.gpus, (td.get_world_size(), args.gpus)
        selfcheck = collective_selfcheck(device, backend)
        if not selfcheck["ok_on_every_rank"]:
            if rank == 0:
                print(json.dumps({"metric": "rows/sec + GB/s (Criteo Categorify+FillMissing+Normalize "
                                            "fit+transform, HBM-resident)", "value": None,
                                  "n_gpus": world, "error": "collective self-check failed: the "
                                  f"{backend} collectives disagree with gloo or raised",
                                  "collective_selfcheck": selfcheck}))
            td.destroy_process_group()
            raise SystemExit(3)

    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K

    n = args.rows
    frame = synth_criteo(n, device, seed=20260923 + 1000 * rank)
    cat_names = [c for c in frame.columns if c.startswith("C")]
    cont_names = [c for c in frame.columns if c.startswith("I")]
    torch.cuda.synchronize()

    tmp = tempfile.mkdtemp(prefix="nvt_bench_")
    wf = build_workflow(cat_names, cont_names, os.path.join(tmp, f"gpu{rank}"))
    ds = nvt.Dataset(frame)

    marks = []  # host clock at (step start, fit returned, transform enqueued): where a slow
                # step spends its time (fit contains the step's one host synchronisation)

    def step():
        t_a = time.perf_counter()
        wf.fit(ds)
        t_b = time.perf_counter()
        out = wf.transform(frame)
        marks.append((t_a, t_b, time.perf_counter()))
        return out

    def barrier():
        if world > 1:
            import torch.distributed as td

            td.barrier()
        torch.cuda.synchronize()

    import gc

    # cold step: a fresh workflow with no cardinality hints (reported, never `value`)
    barrier()
    K.STATS["count_relaunches"] = 0
    cold_alloc0 = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    out = step()
    t_enq = time.perf_counter()
    barrier()
    cold_ms = 1e3 * (time.perf_counter() - t0)
    cold_info = {
        "fit_ms": round(1e3 * (marks[-1][1] - marks[-1][0]), 2),
        "transform_enqueue_ms": round(1e3 * (marks[-1][2] - marks[-1][1]), 2),
        "drain_ms": round(cold_ms - 1e3 * (t_enq - t0), 2),
        "device_allocs": int(torch.cuda.memory_stats(device).get("num_device_alloc", 0) - cold_alloc0),
        "count_relaunches": int(K.STATS["count_relaunches"]),
        "presampled_columns": int(K.STATS["presampled_columns"]),
    }
    # The timed loop keeps the previous step's output frame alive while the next one is
    # produced (`out = step()`), so TWO 14 GB output sets coexist.  If the caching allocator
    # first meets that during the timed region it has to hipMalloc 39 x 360 MB blocks with the
    # GPU busy: one 28-40 ms step (100-180 ms on a box's first process), i.e. +2 ms on the mean
    # of 15-20 steps -- the "outlier" of earlier rounds (profiles/r02_notes.md).  The warm-up
    # therefore runs with the same ownership pattern, and one spare block per output column is
    # cached up front so that this also holds for --warmup 1.
    cold_allocator = os.environ.get("NVT_BENCH_COLD_ALLOCATOR") == "1"  # diagnostic: old warm-up
    if cold_allocator:
        del out
    else:
        spare = [torch.empty_like(t) for _, col in out.items() for t in (col.data, col.valid)
                 if t is not None]
        del spare
    for _ in range(max(args.warmup - 1, 0)):
        if cold_allocator:
            step()
        else:
            out = step()
    gc.collect()
    gc.disable()  # no collector pauses inside the timed region
    # ---- timed region: exactly `steps` steps, no per-kernel instrumentation ----
    barrier()
    del marks[:]
    n_alloc0 = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    n_alloc = torch.cuda.memory_stats(device).get("num_device_alloc", 0) - n_alloc0
    timed_marks = list(marks)
    # ---- second pass, same steps, HIP events on every kernel family (inside the library,
    # on the launch streams): GPU-busy time of the real, overlapped pipeline.
    barrier()
    K.profile_begin()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt_prof = time.perf_counter() - t1
    rep = K.profile_report()
    # ---- third pass: the same steps with the cross-stream overlap switched off (vocabularies
    # ordered inside fit, their streams joined before transform), so that every kernel family
    # is timed running ALONE: these are the per-kernel durations the roofline is computed from
    # (under overlap a bandwidth-bound kernel's own duration stretches while the step shrinks).
    from nvtabular_amd.ops import categorify as _cat

    saved = (K.ASYNC_FINALIZE, _cat.LAZY_FINALIZE, K.COUNT_STREAMS)
    K.ASYNC_FINALIZE, _cat.LAZY_FINALIZE, K.COUNT_STREAMS = False, False, 1
    os.environ["NVT_FINALIZE_SERIAL"] = "1"  # read by the library at every finalize call
    os.environ["NVT_ENCODE_STREAMS"] = "1"   # ... and at every nvt_encode_many call
    step()
    barrier()
    K.profile_begin()
    t2 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt_serial = time.perf_counter() - t2
    prof = K.profile_report()["kernels"]
    K.ASYNC_FINALIZE, _cat.LAZY_FINALIZE, K.COUNT_STREAMS = saved
    os.environ.pop("NVT_FINALIZE_SERIAL", None)
    os.environ.pop("NVT_ENCODE_STREAMS", None)
    gc.enable()
    del out
    # a second hint-less step, now that the library's code objects are loaded and the caching
    # allocator holds the blocks: what a FRESH workflow costs beyond a steady-state refit
    # (presample, conservative path choice, 1024-bucket range path) without the process warm-up
    wf_cold = build_workflow(cat_names, cont_names, os.path.join(tmp, f"gpu{rank}_cold2"))
    barrier()
    t3 = time.perf_counter()
    wf_cold.fit(ds)
    out = wf_cold.transform(frame)
    barrier()
    fresh_ms = 1e3 * (time.perf_counter() - t3)
    cold_info["fresh_workflow_warm_process_ms"] = round(fresh_ms, 2)
    # full-frame parity of the TIMED frame (the state the timed steps left behind: same workflow,
    # same frame), through the size-independent properties of tests/test_gpu_fullsize.py
    full_frame = None
    if world == 1:
        try:
            full_frame = property_checks(wf, [frame], [wf.transform(frame)], cat_names, cont_names)
        except Exception as e:  # never break the line
            full_frame = {"full_frame_ok": False, "error": repr(e)}
    del out, wf_cold
    dist_diag = None
    if world > 1:
        # ONE more fit + transform, outside every timed region, with the exchange's sections timed
        # (device-synchronised marks) and the bytes of every collective form counted: the first
        # RCCL run explains itself -- where the ranks wait, what crosses the links
        from nvtabular_amd import dist as _dd

        barrier()
        _dd.reset_traffic()
        _dd.enable_timing(True)
        t4 = time.perf_counter()
        step()
        barrier()
        diag_ms = 1e3 * (time.perf_counter() - t4)
        dist_diag = {"step_ms_with_synchronised_marks": round(diag_ms, 3),
                     "sections_ms": {k: round(1e3 * v, 3) for k, v in (_dd.TIMING or {}).items()},
                     "collectives": {k: dict(v) for k, v in _dd.TRAFFIC.items()},
                     "bytes_sent_per_rank": sum(v["bytes_sent"] for v in _dd.TRAFFIC.values()),
                     "bytes_received_per_rank": sum(v["bytes_received"] for v in _dd.TRAFFIC.values()),
                     "collective_calls_per_fit": sum(v["calls"] for v in _dd.TRAFFIC.values()),
                     "note": "rank 0's view of one fit + transform outside the timed steps"}
        _dd.enable_timing(False)
    if world > 1:
        import torch.distributed as td

        t = torch.tensor([dt, dt_prof, dt_serial], device=device, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt, dt_prof, dt_serial = (float(v) for v in t.tolist())

    ms_per_step = 1e3 * dt / args.steps
    rows_per_s = world * n * args.steps / dt
    # algorithmic bytes: fit 4 B/value read; transform 4 B read + 8 B written
    C, Kc = len(cat_names), len(cont_names)
    bytes_per_row = (C * 4 + Kc * 4) + (C * 12 + Kc * 12)
    gbs = rows_per_s * bytes_per_row / 1e9

    # Roofline: the kernel FAMILY with the largest summed time among those that stream column
    # data (HIP events around every launch scope, recorded by the library on the launch stream in
    # the third pass; a family = all paths / columns of one operator stage).  achieved = the
    # family's algorithmic bytes per step / its time per step; "launch" = one column's pass.
    roofline = None
    if prof:
        ncols = {"cat": C, "cont": Kc}
        fam = {}
        for scope, (tot_ms, launches, _) in prof.items():
            f = fam.setdefault(family_of(scope), {"ms": 0.0, "launches": 0})
            f["ms"] += tot_ms / args.steps
            if scope != "dense_count_sample":  # (one batched launch per step, not a column pass)
                f["launches"] += launches / args.steps
        traffic = pmc_traffic() if world == 1 and n == 45_000_000 else None
        per_family = {}
        for name, f in fam.items():
            _, bpr, kind = FAMILIES.get(name, ((), 0, "cat"))
            fbytes = bpr * n * ncols[kind]
            per_family[name] = {
                "algorithmic_bytes_per_step": fbytes, "ms_per_step": round(f["ms"], 3),
                "launches_per_step": round(f["launches"], 1),
                "GBps": round(fbytes / (f["ms"] / 1e3) / 1e9, 1) if f["ms"] > 0 else None,
                "frac": round(fbytes / (f["ms"] / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if f["ms"] > 0 else None,
                "hbm_traffic_bytes_per_step": (traffic or {}).get("families", {}).get(name),
            }
        dom = max((k for k, v in per_family.items() if v["algorithmic_bytes_per_step"] > 0),
                  key=lambda k: per_family[k]["ms_per_step"])
        d = per_family[dom]
        launches = max(d["launches_per_step"], 1.0)
        roofline = {
            "bound": "hbm", "kernel": dom, "achieved": d["GBps"], "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": d["frac"],
            "traffic": (d["hbm_traffic_bytes_per_step"] / launches
                        if d["hbm_traffic_bytes_per_step"] else None),
            "traffic_source": (traffic or {}).get("source"),
            "avg_launch_us": round(1e3 * d["ms_per_step"] / launches, 2),
            "launches": int(round(launches * args.steps)),
            "algorithmic_bytes_per_launch": int(d["algorithmic_bytes_per_step"] / launches),
            "per_family": per_family,
            "per_kernel_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in prof.items()},
            "launch_scopes_per_step": round(sum(v[1] for v in prof.values()) / args.steps, 1),
            "measured_in": "third pass: cross-stream overlap off (NVT_ASYNC_FINALIZE=0 "
                           "NVT_LAZY_FINALIZE=0 NVT_COUNT_STREAMS=1 NVT_FINALIZE_SERIAL=1 "
                           "NVT_ENCODE_STREAMS=1), "
                           "every kernel family timed alone; "
                           f"{round(1e3 * dt_serial / args.steps, 3)} ms per step in that mode",
            "overlapped_per_kernel_ms_per_step": {k: round(v[0] / args.steps, 3)
                                                  for k, v in rep["kernels"].items()},
        }
        step_traffic = (traffic or {}).get("step")
    else:
        step_traffic = None

    parity_failed = False
    result = {
        "metric": "rows/sec + GB/s (Criteo Categorify+FillMissing+Normalize fit+transform, HBM-resident)",
        "value": rows_per_s,
        "unit": "rows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "gpu_busy_ms_per_step": round(rep["busy_ms"] / args.steps, 3),
        "profiled_pass_ms_per_step": round(1e3 * dt_prof / args.steps, 3),
        "cold_step_ms": round(cold_ms, 2),
        "cold_step": cold_info,
        # what a user's FIRST fit of a workflow costs in a warm process (no cardinality hints:
        # presample + conservative paths); `value` is the steady-state refit
        "fresh_fit": {"ms_per_step": round(fresh_ms, 2), "rows_per_s": world * n / (fresh_ms / 1e3),
                      "frac_of_hbm_peak": world * n / (fresh_ms / 1e3) * ((26 * 4 + N_CONT * 4) + (26 * 12 + N_CONT * 12))
                      / 1e9 / (HBM_PEAK_GBS * world),
                      "state": "fresh workflow (no hints), process warm"},
        "host_timeline_ms": _host_timeline(timed_marks),
        # hipMalloc calls the caching allocator had to make inside the timed region (0 = the
        # warm-up reached the steady-state footprint)
        "device_allocs_in_timed_region": int(n_alloc),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32 keys -> int64 labels; fp64 moments/normalize",
        "data": "synthetic",
        "config": {
            "workload": "synthetic Criteo-day0: 26 int32 categorical (Criteo-1TB cardinalities, "
                        "Zipf) + 13 int32 continuous, nulls as Arrow bitmaps; "
                        "Categorify + FillMissing + Normalize, fit + transform",
            "rows_per_gpu": n,
            "algorithmic_bytes_per_row": bytes_per_row,
            "tie_break": "value (count desc, value asc)",
            "artifacts": "deferred (no parquet I/O inside the timed region)",
            "state": "steady-state refit (cardinality hints from the warm-up steps; "
                     "cold_step_ms = first step of a fresh workflow)",
            "collective_backend": backend,
        },
        "algorithmic_GBps": gbs,
        "frac_of_hbm_peak": gbs / (HBM_PEAK_GBS * world),
        "algorithmic_bytes_per_step": bytes_per_row * n,
        "step_traffic_bytes": step_traffic,  # HBM bytes per step from the committed PMC passes
        "roofline": roofline,
    }
    if rank == 0 and not args.no_cpu_baseline:
        # (rank 0 only, also when world > 1: its own shard, a single-rank fit for the parity leg)
        from nvtabular_amd import dist as _dist

        sample = min(args.cpu_sample, n) // 8 * 8
        result["cpu_baseline"], oracle_out = cpu_baseline(frame, cat_names, cont_names, sample, tmp)
        with _dist.local_only():
            result["parity"] = parity_check(frame, cat_names, cont_names, sample, oracle_out, tmp)
        if full_frame is not None:
            result["parity"].update({"full_frame_ok": full_frame.get("full_frame_ok"),
                                     "full_frame": full_frame})
        result["parity"]["tie_break_note"] = (
            "labels equal the reference up to permutations inside equal-count blocks "
            "(tie_break='value': count desc, value asc; the reference's second sort_values is "
            "pandas' unstable default); tie_break='reference' reproduces the literal order on the "
            "host: reference_tie_break_step_ms")
        try:
            with _dist.local_only():   # (rank 0 alone: its peers are not in this fit's collectives)
                result["reference_tie_break_step_ms"] = round(
                    reference_tie_break_step_ms(frame, cat_names, cont_names, tmp), 1)
        except Exception as e:
            result["reference_tie_break_step_ms"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_extra and not args.no_cpu_baseline:
        # entries beside the headline number (never `value`); none of them may break the line
        try:
            result["eager_artifacts_step_ms"] = round(
                eager_artifacts_step_ms(frame, cat_names, cont_names, tmp), 1)
        except Exception as e:
            result["eager_artifacts_step_ms"] = {"error": repr(e)}
        del frame, ds, wf
        torch.cuda.empty_cache()
        extras = {}
        for key, fn in (("cfg3_multipartition", lambda: extra_multipart(device, tmp, n, args.multipart,
                                                                        single_ms=ms_per_step)),
                        ("cfg2_dense_ids", lambda: extra_dense_ids(device, tmp, n, single_ms=ms_per_step)),
                        ("cfg4_te_joingroupby", lambda: extra_cfg4(device, tmp, args.cfg4_rows, 1_000_000)),
                        ("cfg4_multipartition", lambda: extra_cfg4_multipart(
                            device, tmp, args.cfg4_part_rows, args.cfg4_parts,
                            single=(extras.get("cfg4_te_joingroupby") or {}).get("rows_per_s"))),
                        ("cfg3_highcard_columns", lambda: extra_cfg3(device, tmp, n)),
                        ("cfg5_multihot", lambda: extra_cfg5(device, tmp)),
                        ("end_to_end", lambda: extra_end_to_end(device, tmp, n))):
            if args.only_extra and key not in args.only_extra.split(","):
                continue
            try:
                extras[key] = fn()
            except Exception as e:
                extras[key] = {"error": repr(e)}
            torch.cuda.empty_cache()
        result["extra_configs"] = extras
    if world > 1:
        result["collective_selfcheck"] = selfcheck
        result["dist_breakdown"] = dist_diag
        if not args.no_extra:
            # (every rank: the fit's collectives need all of them)
            for name in ("frame", "ds", "wf"):
                locals().pop(name, None)
            torch.cuda.empty_cache()
            barrier()   # (rank 0 comes out of its single-rank legs here)
            try:
                result["dist_cfg3_multipartition"] = dist_multipart(device, tmp, n, args.multipart, rank, world,
                                                                    barrier)
            except Exception as e:  # noqa: BLE001
                result["dist_cfg3_multipartition"] = {"error": repr(e)}
    if rank == 0:
        from nvtabular_amd import dist as _d

        if _d.TIMING:  # NVT_DIST_TIMING=1 (diagnostic, device-synchronised sections of the merge)
            result["dist_timing_ms_total"] = {k: round(1e3 * v, 2) for k, v in _d.TIMING.items()}
        if world > 1:  # which exchange / ordering paths the fits of this run took (rank 0's counters)
            result["dist_stats"] = dict(_d.STATS)
        # parity is loud: one top-level verdict over every leg of the line (the nested legs keep
        # their details) and a non-zero exit when any of them failed
        if "parity" in result or "extra_configs" in result:
            ok, prow, failed = parity_verdicts(result)
            result["parity_ok"] = ok
            result["parity_rows"] = prow
            result["parity_failed"] = failed
            parity_failed = not ok
        print(json.dumps(result))
    if world > 1:
        import torch.distributed as td

        td.destroy_process_group()
    if parity_failed:
        print("bench.py: PARITY FAILED: " + ", ".join(result["parity_failed"]), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
