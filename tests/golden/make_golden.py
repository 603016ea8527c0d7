"""Generates tests/golden/hotpath_v1.npz -- seeded inputs and the outputs the CPU oracle
(``oracle/nvt_oracle.py``, the pandas restatement of the reference pinned to the reference's
own golden vectors in tests/test_oracle_golden.py) produces for them.

The reference itself cannot be imported in this image (merlin-core / dask are absent), so
these fixtures are *oracle* outputs frozen at a known-good commit: the CPU suite checks the
oracle still reproduces them (drift guard), the GPU suite checks the HIP path against them
without running the oracle at all.

    python tests/golden/make_golden.py         # rewrites hotpath_v1.npz
"""
import os
import sys
import tempfile

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle as O  # noqa: E402


def make_inputs():
    rng = np.random.default_rng(20260923)
    n = 20_000
    raw = np.minimum(rng.zipf(1.3, n), 700).astype(np.int64)
    cat_a = ((raw * 2654435761) % (2**31 - 1) - 2**30).astype(np.int32)
    cat_b = rng.integers(-(2**62), 2**62, size=n, dtype=np.int64)
    cat_b = cat_b[rng.integers(0, 3000, size=n)]  # 3000 distinct-ish int64 keys, uniform
    null_a = rng.random(n) < 0.1
    null_b = rng.random(n) < 0.02
    cont_x = rng.normal(3.0, 20.0, size=n).astype(np.float32)
    cont_y = rng.exponential(5.0, size=n)
    null_x = rng.random(n) < 0.15
    label = (rng.random(n) < 0.3).astype(np.float64)
    return dict(cat_a=cat_a, cat_b=cat_b, null_a=null_a, null_b=null_b, cont_x=cont_x,
                cont_y=cont_y, null_x=null_x, label=label)


def oracle_frame(inp):
    a = pd.Series(inp["cat_a"].astype("float64"))
    a[inp["null_a"]] = np.nan
    b = pd.array(inp["cat_b"], dtype="Int64")
    b[inp["null_b"]] = pd.NA
    x = pd.Series(inp["cont_x"].copy())
    x[inp["null_x"]] = np.nan
    return pd.DataFrame({"cat_a": a, "cat_b": pd.Series(b), "cont_x": x, "cont_y": inp["cont_y"],
                         "label": inp["label"]})


def expected(inp):
    df = oracle_frame(inp)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        # Categorify, default options (stable tie rule)
        paths = O.categorify_fit([df], ["cat_a", "cat_b"], tmp + "/c0", tie_break="stable")
        enc = O.categorify_transform(df, ["cat_a", "cat_b"], paths)
        out["enc_a"] = enc["cat_a"].to_numpy().astype(np.int64)
        out["enc_b"] = enc["cat_b"].to_numpy().astype(np.int64)
        va = pd.read_parquet(paths["cat_a"])
        out["vocab_a"] = va["cat_a"].to_numpy().astype(np.int64)
        out["vocab_a_size"] = va["cat_a_size"].to_numpy().astype(np.int64)
        # freq_threshold + hashed OOV buckets
        p2 = O.categorify_fit([df], ["cat_a"], tmp + "/c1", tie_break="stable", freq_threshold=3,
                              num_buckets=11)
        out["enc_a_ft3_nb11"] = (
            O.categorify_transform(df, ["cat_a"], p2, num_buckets=11)["cat_a"].to_numpy().astype(np.int64)
        )
        # JoinGroupby(count, mean, sum) on cat_a over cont_y
        cats = O.join_groupby_fit([df], [["cat_a"]], ["cont_y"], ["count", "mean", "sum"], tmp + "/jg")
        jg = O.join_groupby_transform(df, [["cat_a"]], cats)
        for c in sorted(jg.columns):
            out["jg_" + c] = jg[c].to_numpy().astype(np.float64)
    # HashBucket
    hb = O.hash_bucket_op(df[["cat_b"]].fillna(0).astype("int64"), 97, ["cat_b"])
    out["hash_b_97"] = hb["cat_b"].to_numpy().astype(np.int64)
    # FillMissing >> Normalize
    filled = O.fill_missing(df.copy(), ["cont_x", "cont_y"], 0)
    mom = O.custom_moments([filled], ["cont_x", "cont_y"])
    means, stds = mom["mean"].to_dict(), mom["std"].to_dict()
    out["means"] = np.array([means["cont_x"], means["cont_y"]], dtype=np.float64)
    out["stds"] = np.array([stds["cont_x"], stds["cont_y"]], dtype=np.float64)
    nz = O.normalize_transform(filled, ["cont_x", "cont_y"], means, stds)
    out["norm_x"] = nz["cont_x"].to_numpy().astype(np.float64)
    out["norm_y"] = nz["cont_y"].to_numpy().astype(np.float64)
    # FillMissing >> Clip(min_value=0) >> LogOp
    cl = O.logop_transform(O.clip_transform(filled, ["cont_x"], min_value=0), ["cont_x"])
    out["log_x"] = cl["cont_x"].to_numpy().astype(np.float32)
    return out


# ---- second fixture: ONE int32 key column without nulls, enough rows for the engine's sort path
# of the single-key groupby (JoinGroupby / TargetEncoding, >= 32768 rows) --------------------------
def make_inputs_groupby():
    rng = np.random.default_rng(20260924)
    n = 36_000
    ids = rng.integers(-(2**31), 2**31 - 1, 6_000, dtype=np.int64).astype(np.int32)
    key = ids[(rng.random(n) ** 3 * ids.size).astype(np.int64)]      # skewed frequencies
    x = rng.normal(1.0, 4.0, size=n)
    y = (rng.random(n) < 0.25).astype(np.float32)
    null_x = rng.random(n) < 0.1
    return dict(key=key, x=x, y=y, null_x=null_x)


def oracle_frame_groupby(inp):
    x = pd.Series(inp["x"].copy())
    x[inp["null_x"]] = np.nan
    return pd.DataFrame({"k": inp["key"], "x": x, "y": inp["y"]})


def expected_groupby(inp):
    df = oracle_frame_groupby(inp)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        stats = ["count", "sum", "mean", "std", "min", "max"]
        cats = O.join_groupby_fit([df.copy()], ["k"], ["x"], stats, tmp + "/jg")
        jg = O.join_groupby_transform(df.copy(), ["k"], cats)
        for c in sorted(jg.columns):
            if c not in ("k", "x", "y"):
                out["jg_" + c] = jg[c].to_numpy()  # (count int32, mean / std float32, the rest float64)
        for kfold, seed in ((5, 42), (1, None)):
            st, means = O.target_encoding_fit([df.copy()], ["k"], ["y"], tmp + f"/te{kfold}", kfold=kfold,
                                              fold_seed=seed)
            te = O.target_encoding_transform(df[["k", "y"]].copy(), ["k"], ["y"], st, means, kfold=kfold,
                                             fold_seed=seed, p_smooth=20)
            out[f"te_k{kfold}"] = te["TE_k_y"].to_numpy().astype(np.float32)
    return out


def main():
    inp = make_inputs()
    exp = expected(inp)
    np.savez_compressed(os.path.join(HERE, "hotpath_v1.npz"),
                        **{"in_" + k: v for k, v in inp.items()},
                        **{"out_" + k: v for k, v in exp.items()})
    print({k: (v.shape, v.dtype) for k, v in exp.items()})
    inp = make_inputs_groupby()
    exp = expected_groupby(inp)
    np.savez_compressed(os.path.join(HERE, "groupby_v1.npz"),
                        **{"in_" + k: v for k, v in inp.items()},
                        **{"out_" + k: v for k, v in exp.items()})
    print({k: (v.shape, v.dtype) for k, v in exp.items()})


if __name__ == "__main__":
    main()
