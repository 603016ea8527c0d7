"""Sort path of the single-key groupby (nvt_sgb_sort + nvt_sgb_regroup + nvt_sgb_reduce, the
flat index and the fused transforms nvt_flat_lookup_gather / nvt_flat_lookup_te):
JoinGroupby / TargetEncoding fit and transform on ONE int32 key column.  Checked against
pandas groupby at the kernel level and against the oracle's restatement of the reference
(categorify.py:955-1137, join_groupby.py:175-217, target_encoding.py:254-384) end to end."""
import numpy as np
import pandas as pd
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


def _skewed_keys(rng, n, card, lo=-(2**31), hi=2**31 - 1):
    ids = rng.integers(lo, hi, card, dtype=np.int64).astype(np.int32)
    pick = (rng.random(n) ** 3 * card).astype(np.int64)
    return ids[pick]


@pytest.mark.parametrize("kfold", [1, 5, 16])
def test_sgb_aggregate_vs_pandas(kfold):
    from nvtabular_amd import kernels as K
    from nvtabular_amd.device import pack_bitmap_device

    rng = np.random.default_rng(5 + kfold)
    n, card = 200_003, 30_000
    key = _skewed_keys(rng, n, card)
    key[:3] = [np.iinfo(np.int32).min, np.iinfo(np.int32).max, 0]
    x = rng.normal(size=n)
    x[rng.random(n) < 0.05] = np.nan
    y = rng.random(n).astype(np.float32)
    yvalid = rng.random(n) < 0.9
    fold = rng.integers(0, kfold, n).astype(np.uint8) if kfold > 1 else None
    dev = "cuda"
    tk = torch.from_numpy(key).to(dev)
    vals = [torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)]
    vvalid = [None, pack_bitmap_device(torch.from_numpy(yvalid).to(dev))]
    tf = torch.from_numpy(fold).to(dev) if fold is not None else None
    minmax = kfold == 1
    for hint in (0, 100):  # 100: too small -> the exact-size relaunch
        comp = K.sorted_groupby(tk, tf, kfold, vals, vvalid, sumsq=minmax, minmax=minmax, cap_hint=hint)
        df = pd.DataFrame({"k": key, "x": x, "y": np.where(yvalid, y.astype(np.float64), np.nan)})
        gb = df.groupby("k", sort=True)
        exp_keys = np.array(sorted(set(key.tolist())), dtype=np.int64)
        assert comp["n"] == exp_keys.size
        np.testing.assert_array_equal(comp["keys"][0].cpu().numpy(), exp_keys)
        np.testing.assert_array_equal(comp["keys32"].cpu().numpy(), exp_keys.astype(np.int32))
        np.testing.assert_array_equal(comp["size"].cpu().numpy(), gb.size().to_numpy())
        for j, c in enumerate(["x", "y"]):
            np.testing.assert_allclose(comp["sum"][j].cpu().numpy(), gb[c].sum().to_numpy(),
                                       rtol=1e-11, atol=1e-11)
        if minmax:
            for j, c in enumerate(["x", "y"]):
                np.testing.assert_allclose(comp["sumsq"][j].cpu().numpy(),
                                           (df[c] ** 2).groupby(df["k"]).sum().to_numpy(), rtol=1e-11, atol=1e-11)
                np.testing.assert_array_equal(comp["min"][j].cpu().numpy(), gb[c].min().to_numpy())
                np.testing.assert_array_equal(comp["max"][j].cpu().numpy(), gb[c].max().to_numpy())
        if kfold > 1:
            f = comp["fold"]
            df["f"] = fold
            pos = np.searchsorted(exp_keys, key.astype(np.int64)) * kfold + fold
            size = np.bincount(pos, minlength=exp_keys.size * kfold)
            np.testing.assert_array_equal(f["size"].cpu().numpy(), size)
            for j, c in enumerate(["x", "y"]):
                v = df[c].to_numpy()
                s = np.bincount(pos[~np.isnan(v)], weights=v[~np.isnan(v)], minlength=size.size)
                np.testing.assert_allclose(f["sum"][j].cpu().numpy(), s, rtol=1e-11, atol=1e-11)


def test_flat_index_lookup_and_clustered_fallback():
    from nvtabular_amd import kernels as K
    from nvtabular_amd.device import pack_bitmap_device

    rng = np.random.default_rng(2)
    keys = np.unique(rng.integers(-(2**31), 2**31 - 1, 300_000, dtype=np.int64).astype(np.int32))
    keys[0] = np.iinfo(np.int32).min  # (still ascending: the smallest int32)
    dev = "cuda"
    idx = K.FlatIndex(torch.from_numpy(keys).to(dev))
    assert idx.ok()
    probe = np.concatenate([keys[::3], rng.integers(-(2**31), 2**31 - 1, 100_000, dtype=np.int64).astype(np.int32)])
    pos = np.searchsorted(keys, probe)
    pos_c = np.minimum(pos, keys.size - 1)
    exp = np.where(keys[pos_c] == probe, pos_c, -1)
    got = idx.lookup([torch.from_numpy(probe).to(dev)], [None]).cpu().numpy()
    np.testing.assert_array_equal(got, exp)
    # int64 column (values outside int32 are unseen), validity bitmap (null key -> -1)
    p64 = probe.astype(np.int64)
    p64[:10] += 2**40
    valid = rng.random(p64.size) < 0.8
    got = idx.lookup([torch.from_numpy(p64).to(dev)],
                     [pack_bitmap_device(torch.from_numpy(valid).to(dev))]).cpu().numpy()
    exp64 = exp.copy()
    exp64[:10] = -1
    exp64[~valid] = -1
    np.testing.assert_array_equal(got, exp64)
    # without the smallest int32 in the list it is unseen
    idx2 = K.FlatIndex(torch.from_numpy(keys[1:].copy()).to(dev))
    q = torch.tensor([np.iinfo(np.int32).min, int(keys[1]), int(keys[-1])], dtype=torch.int32, device=dev)
    assert idx2.lookup([q], [None]).tolist() == [-1, 0, keys.size - 2]
    # dense ids + one far outlier: every key has the same home slot -> not ok (hashed index instead)
    clustered = np.concatenate([np.arange(100_000, dtype=np.int32), np.array([2**31 - 5], dtype=np.int32)])
    idx3 = K.FlatIndex(torch.from_numpy(clustered).to(dev))
    assert idx3.table is not None and not idx3.ok()   # (the table is laid out on first use)
    # ... and still finds every key (galloping steps inside the long run)
    q = torch.tensor([0, 77_777, 99_999, 100_000, 2**31 - 5], dtype=torch.int32, device=dev)
    assert idx3.lookup([q], [None]).tolist() == [0, 77_777, 99_999, -1, 100_000]


def _frames(n, card, seed, key="int32"):
    rng = np.random.default_rng(seed)
    k = _skewed_keys(rng, n, card, 0, 2**31 - 1)
    if key == "int64":  # an int64 column whose keys span less than 2^32, far from zero
        k = k.astype(np.int64) * 2 - 7_000_000_000_000
    df = pd.DataFrame({
        "k": k,
        "x": rng.normal(size=n),
        "y": (rng.random(n) < 0.3).astype("float32"),
    })
    df.loc[rng.random(n) < 0.05, "x"] = np.nan
    return df


def _split(df, nparts):
    cuts = np.linspace(0, len(df), nparts + 1).astype(int)
    cuts[1:-1] += np.arange(1, nparts) * 1777  # uneven partitions
    return [df.iloc[a:b].reset_index(drop=True) for a, b in zip(cuts[:-1], cuts[1:])]


@pytest.mark.parametrize("key", ["int32", "int64"])
@pytest.mark.parametrize("nparts", [1, 2, 3])
def test_joingroupby_sorted_path_vs_oracle(tmp_path, nparts, key):
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K
    from nvtabular_amd import ops

    df = _frames(120_000 if nparts < 3 else 180_000, 9_000, 21, key)
    parts = _split(df, nparts)
    stats = ["count", "sum", "mean", "std", "var", "min", "max"]
    jg = ops.JoinGroupby(out_path=str(tmp_path / "g"), stats=stats, cont_cols=["x", "y"])
    wf = nvt.Workflow(["k"] >> jg).fit(nvt.Dataset(parts))
    index = jg._device_stats["k"].index
    # sorted groups + flat index however many partitions: the partitions' key-ordered groups are
    # merged by the merge-path kernel (int64 keys: images re-based to the first partition's offset)
    assert isinstance(index, K.FlatIndex)
    got = wf.transform(nvt.Dataset(parts)).to_ddf().compute()
    cats = O.join_groupby_fit([p.copy() for p in parts], ["k"], ["x", "y"], stats, str(tmp_path / "c"))
    exp = O.join_groupby_transform(df.copy(), ["k"], cats)
    assert sorted(got.columns) == sorted(exp.columns)
    for c in exp.columns:
        if c.endswith("_count"):
            np.testing.assert_array_equal(got[c].to_numpy(), exp[c].to_numpy())
        else:
            np.testing.assert_allclose(got[c].to_numpy().astype("float64"),
                                       exp[c].to_numpy().astype("float64"), rtol=2e-5, atol=1e-6, err_msg=c)
    # the artifact lists the same groups with the same statistics
    a = pd.read_parquet(jg.categories["k"]).sort_values("k").reset_index(drop=True)
    b = pd.read_parquet(cats["k"]).sort_values("k").reset_index(drop=True)
    assert list(a["k"]) == list(b["k"]) and a["k"].dtype == b["k"].dtype
    np.testing.assert_array_equal(a["k_count"].to_numpy(), b["k_count"].to_numpy())
    np.testing.assert_allclose(a["k_x_sum"].to_numpy(), b["k_x_sum"].to_numpy(), rtol=1e-9, atol=1e-9)
    # unseen keys and nulls at transform time
    new = pd.DataFrame({"k": pd.array([int(df["k"][0]), 2**31 - 2, None], dtype="Int64").astype("float64"),
                        "x": [0.0, 0.0, 0.0], "y": np.zeros(3, dtype="float32")})
    jg2 = ops.JoinGroupby(out_path=str(tmp_path / "g2"), stats=["sum"], cont_cols=["x"])
    wf2 = nvt.Workflow(["k"] >> jg2).fit(nvt.Dataset(parts))
    o2 = wf2.transform(nvt.Dataset(new)).to_ddf().compute()
    assert np.isfinite(o2["k_x_sum"][0]) and np.isnan(o2["k_x_sum"][1]) and np.isnan(o2["k_x_sum"][2])


@pytest.mark.parametrize("key", ["int32", "int64"])
@pytest.mark.parametrize("kfold,fold_seed", [(1, None), (5, 42), (3, None)])
@pytest.mark.parametrize("nparts", [1, 2, 3])
def test_target_encoding_sorted_path_vs_oracle(tmp_path, kfold, fold_seed, nparts, key):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops
    from nvtabular_amd.ops.target_encoding import _FoldDense

    df = _frames(100_000 if nparts < 3 else 150_000, 7_000, 33, key)
    parts = _split(df, nparts)
    te = ops.TargetEncoding(["y", "x"], out_path=str(tmp_path / "g"), kfold=kfold, fold_seed=fold_seed,
                            p_smooth=20)
    wf = nvt.Workflow(["k"] >> te).fit(nvt.Dataset(parts))
    if kfold > 1:  # dense per-(group, fold) statistics survive the partition merge
        assert isinstance(te._device_stats["__fold___k"], _FoldDense)
    got = wf.transform(nvt.Dataset(parts)).to_ddf().compute()
    oparts = [p.copy() for p in parts]
    stats, means = O.target_encoding_fit(oparts, ["k"], ["y", "x"], str(tmp_path / "c"), kfold=kfold,
                                         fold_seed=fold_seed)
    exp = pd.concat([
        O.target_encoding_transform(p[["k", "y", "x"]].copy(), ["k"], ["y", "x"], stats, means,
                                    kfold=kfold, fold_seed=fold_seed, p_smooth=20)
        for p in parts], ignore_index=True)
    assert list(got.columns) == list(exp.columns)
    for c in exp.columns:
        assert got[c].dtype == np.float32
        np.testing.assert_allclose(got[c].to_numpy(), exp[c].to_numpy(), rtol=1e-5, atol=1e-6, err_msg=c)
    if kfold > 1:  # the [fold, key] artifact: same groups, counts and sums as the reference's
        a = pd.read_parquet(te.stats["__fold___k"]).sort_values(["__fold__", "k"]).reset_index(drop=True)
        b = pd.read_parquet(stats["__fold___k"]).sort_values(["__fold__", "k"]).reset_index(drop=True)
        assert list(a.columns) == list(b.columns)
        for c in b.columns:
            np.testing.assert_allclose(a[c].to_numpy().astype("float64"), b[c].to_numpy().astype("float64"),
                                       rtol=1e-9, atol=1e-9, err_msg=c)
    # a frame the fit never saw: unseen keys and unseen (fold, key) pairs fall back to the mean
    other = _frames(50_000, 20_000, 77, key)
    got2 = wf.transform(nvt.Dataset(other)).to_ddf().compute()
    exp2 = O.target_encoding_transform(other[["k", "y", "x"]].copy(), ["k"], ["y", "x"], stats, means,
                                       kfold=kfold, fold_seed=fold_seed, p_smooth=20)
    for c in exp2.columns:
        np.testing.assert_allclose(got2[c].to_numpy(), exp2[c].to_numpy(), rtol=1e-5, atol=1e-6, err_msg=c)


def test_sorted_path_save_load_roundtrip(tmp_path):
    """Workflow.save -> load rebuilds the lookup from the parquet artifacts (hashed index, sparse
    fold table): same output as the dense statistics of the fit."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    df = _frames(80_000, 5_000, 4)
    te = ["k"] >> ops.TargetEncoding("y", out_path=str(tmp_path / "te"), kfold=5, p_smooth=20)
    jg = ["k"] >> ops.JoinGroupby(out_path=str(tmp_path / "jg"), stats=["count", "mean"], cont_cols=["x"])
    wf = nvt.Workflow(te + jg).fit(nvt.Dataset(df))
    a = wf.transform(nvt.Dataset(df)).to_ddf().compute()
    wf.save(str(tmp_path / "wf"))
    b = nvt.Workflow.load(str(tmp_path / "wf")).transform(nvt.Dataset(df)).to_ddf().compute()
    assert list(a.columns) == list(b.columns)
    for c in a.columns:
        np.testing.assert_allclose(a[c].to_numpy().astype("float64"), b[c].to_numpy().astype("float64"),
                                   rtol=1e-6, atol=1e-7, err_msg=c)


@pytest.mark.parametrize("order", ["te_jg", "jg_te"])
def test_aggregates_of_one_pass_share_sort_and_groups(tmp_path, order):
    """TargetEncoding and JoinGroupby on the same column in one workflow: the second aggregate
    reuses the sorted words (and, after TargetEncoding, the group ids regrouped with its folds).
    Same results as each operator fitted alone (the oracle)."""
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K
    from nvtabular_amd import ops

    df = _frames(90_000, 6_000, 55)
    stats = ["count", "sum", "mean", "std"]
    te = ["k"] >> ops.TargetEncoding("y", out_path=str(tmp_path / "te"), kfold=5, fold_seed=42, p_smooth=20)
    jg = ["k"] >> ops.JoinGroupby(out_path=str(tmp_path / "jg"), stats=stats, cont_cols=["x"])
    sorts = []
    real = K._lib.load().nvt_sgb_sort

    wf = nvt.Workflow(te + jg if order == "te_jg" else jg + te)
    calls = {"sort": 0}
    orig = K.sorted_groupby

    def counting(*a, **kw):
        before = K.current_pass_memo() is not None and len(K.current_pass_memo())
        out = orig(*a, **kw)
        calls["sort"] += int(K.current_pass_memo() is not None and len(K.current_pass_memo()) > before)
        return out

    K.sorted_groupby = counting
    try:
        import nvtabular_amd.ops._groupby as G
        wf.fit(nvt.Dataset(df))
    finally:
        K.sorted_groupby = orig
    got = wf.transform(nvt.Dataset(df)).to_ddf().compute()
    cats = O.join_groupby_fit([df.copy()], ["k"], ["x"], stats, str(tmp_path / "c"))
    exp_j = O.join_groupby_transform(df.copy(), ["k"], cats)
    st, means = O.target_encoding_fit([df.copy()], ["k"], ["y"], str(tmp_path / "c2"), kfold=5, fold_seed=42)
    exp_t = O.target_encoding_transform(df[["k", "y"]].copy(), ["k"], ["y"], st, means, kfold=5,
                                        fold_seed=42, p_smooth=20)
    for exp in (exp_j, exp_t):
        for c in exp.columns:
            if c in ("k", "x", "y"):
                continue
            np.testing.assert_allclose(got[c].to_numpy().astype("float64"), exp[c].to_numpy().astype("float64"),
                                       rtol=2e-5, atol=1e-6, err_msg=c)
    assert (got["k_count"].to_numpy() == exp_j["k_count"].to_numpy()).all()


def test_int64_keys_spanning_more_than_32_bits_keep_the_hash_tables(tmp_path):
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K
    from nvtabular_amd import ops

    df = _frames(60_000, 3_000, 8)
    df["k"] = df["k"].astype(np.int64) * 1_000_003  # spans ~2^51
    jg = ops.JoinGroupby(out_path=str(tmp_path / "g"), stats=["count", "sum"], cont_cols=["x"])
    wf = nvt.Workflow(["k"] >> jg).fit(nvt.Dataset(df))
    assert isinstance(jg._device_stats["k"].index, K.GroupbyTable)
    got = wf.transform(nvt.Dataset(df)).to_ddf().compute()
    cats = O.join_groupby_fit([df.copy()], ["k"], ["x"], ["count", "sum"], str(tmp_path / "c"))
    exp = O.join_groupby_transform(df.copy(), ["k"], cats)
    np.testing.assert_array_equal(got["k_count"].to_numpy(), exp["k_count"].to_numpy())
    np.testing.assert_allclose(got["k_x_sum"].to_numpy(), exp["k_x_sum"].to_numpy(), rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("shape", ["smaller_min_fits", "smaller_min_rebase_both", "union_wider_than_32_bits"])
@pytest.mark.parametrize("op", ["join", "te"])
def test_int64_partitions_with_different_key_offsets(tmp_path, shape, op):
    """every partition of an int64 key column picks its own 32-bit image
    offset (its smallest key).  A later partition whose smallest key lies BELOW the first
    partition's must neither lose the earlier partitions' statistics nor count itself twice:
    images that fit the accumulated offset are re-based onto it, a union narrower than 2^32 gets
    a common offset, a wider union falls back to the hash tables (accumulated groups demoted
    once, the partition's rows hashed once).  Reference semantics: the partition tree of
    categorify.py:955-1137 under join_groupby.py:140-173 / target_encoding.py:171-214."""
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K
    from nvtabular_amd import ops

    rng = np.random.default_rng(5)
    base = 9_000_000_000_000
    n = 60_000

    def part(lo, hi, seed):
        r = np.random.default_rng(seed)
        ids = r.integers(lo, hi, 4_000, dtype=np.int64)
        k = ids[(r.random(n) ** 2 * ids.size).astype(np.int64)] + base
        return pd.DataFrame({"k": k, "x": r.normal(size=n), "y": (r.random(n) < 0.3).astype("float32")})

    if shape == "smaller_min_fits":       # partition 2 reaches 1000 keys below partition 1's minimum
        parts = [part(5_000, 2_000_000, 1), part(4_000, 1_500_000, 2), part(4_500, 2_500_000, 3)]
    elif shape == "smaller_min_rebase_both":  # partition 2 lies 3e9 below: only a common offset fits both
        parts = [part(3_000_000_000, 3_000_900_000, 1), part(0, 900_000, 2), part(1_000_000_000, 3_000_500_000, 3)]
    else:                                  # union of the partitions spans 6e9 >= 2^32
        parts = [part(6_000_000_000, 6_000_900_000, 1), part(0, 900_000, 2), part(10, 6_000_000_500, 3)]
    # shared keys between partitions so that a double count / a lost partition shows
    parts[1] = pd.concat([parts[1], parts[0].iloc[:500].assign(k=parts[1]["k"].iloc[:500].to_numpy())],
                         ignore_index=True)
    df = pd.concat(parts, ignore_index=True)
    if op == "join":
        stats = ["count", "sum", "min", "max"]
        jg = ops.JoinGroupby(out_path=str(tmp_path / "g"), stats=stats, cont_cols=["x"])
        wf = nvt.Workflow(["k"] >> jg).fit(nvt.Dataset(parts))
        index = jg._device_stats["k"].index
        assert isinstance(index, K.GroupbyTable if shape == "union_wider_than_32_bits" else K.FlatIndex)
        got = wf.transform(nvt.Dataset(parts)).to_ddf().compute()
        cats = O.join_groupby_fit([p.copy() for p in parts], ["k"], ["x"], stats, str(tmp_path / "c"))
        exp = O.join_groupby_transform(df.copy(), ["k"], cats)
        np.testing.assert_array_equal(got["k_count"].to_numpy(), exp["k_count"].to_numpy())
        for c in ("k_x_sum", "k_x_min", "k_x_max"):
            np.testing.assert_allclose(got[c].to_numpy(), exp[c].to_numpy(), rtol=1e-9, atol=1e-9, err_msg=c)
    else:
        te = ops.TargetEncoding(["y"], out_path=str(tmp_path / "g"), kfold=3, fold_seed=42, p_smooth=20)
        wf = nvt.Workflow(["k"] >> te).fit(nvt.Dataset(parts))
        got = wf.transform(nvt.Dataset(parts)).to_ddf().compute()
        stats, means = O.target_encoding_fit([p.copy() for p in parts], ["k"], ["y"], str(tmp_path / "c"),
                                             kfold=3, fold_seed=42)
        exp = pd.concat([O.target_encoding_transform(p[["k", "y"]].copy(), ["k"], ["y"], stats, means,
                                                     kfold=3, fold_seed=42, p_smooth=20)
                         for p in parts], ignore_index=True)
        for c in exp.columns:
            np.testing.assert_allclose(got[c].to_numpy(), exp[c].to_numpy(), rtol=1e-5, atol=1e-6, err_msg=c)


@pytest.mark.parametrize("n", [32768, 32769, 34816, 100_001])
@pytest.mark.parametrize("shape", ["one_group", "all_distinct", "two_hot", "runs"])
def test_sgb_edge_shapes(n, shape):
    """Tile / wave boundaries of the run-length pass and the segmented reduction: sizes around
    the 2048-word tile, a single group, all keys distinct, two giant groups, sorted input."""
    from nvtabular_amd import kernels as K

    rng = np.random.default_rng(n)
    if shape == "one_group":
        key = np.full(n, -5, dtype=np.int32)
    elif shape == "all_distinct":
        key = rng.permutation(n).astype(np.int32) - n // 2
    elif shape == "two_hot":
        key = np.where(rng.random(n) < 0.5, np.int32(7), np.int32(2**31 - 1)).astype(np.int32)
    else:
        key = np.sort(rng.integers(0, n // 3, n)).astype(np.int32)
    kfold = 3
    fold = rng.integers(0, kfold, n).astype(np.uint8)
    y = rng.normal(size=n)
    tk, tf, ty = (torch.from_numpy(a).cuda() for a in (key, fold, y))
    for kf, f in ((1, None), (kfold, tf)):
        comp = K.sorted_groupby(tk, f, kf, [ty], [None], sumsq=kf == 1, minmax=kf == 1, te_records=True)
        uk, inv = np.unique(key, return_inverse=True)
        np.testing.assert_array_equal(comp["keys"][0].cpu().numpy(), uk.astype(np.int64))
        np.testing.assert_array_equal(comp["size"].cpu().numpy(), np.bincount(inv, minlength=uk.size))
        np.testing.assert_allclose(comp["sum"][0].cpu().numpy(), np.bincount(inv, weights=y, minlength=uk.size),
                                   rtol=1e-10, atol=1e-9)
        if kf == 1:
            mn = np.full(uk.size, np.inf)
            np.minimum.at(mn, inv, y)
            np.testing.assert_array_equal(comp["min"][0].cpu().numpy(), mn)
        else:
            pos = inv * kfold + fold
            np.testing.assert_array_equal(comp["fold"]["size"].cpu().numpy(),
                                          np.bincount(pos, minlength=uk.size * kfold))
            np.testing.assert_allclose(comp["fold"]["sum"][0].cpu().numpy(),
                                       np.bincount(pos, weights=y, minlength=uk.size * kfold), rtol=1e-10, atol=1e-9)
            rec = comp["fold"]["records"][0].cpu().numpy()
            np.testing.assert_allclose(rec[:, 0], comp["sum"][0].cpu().numpy(), rtol=0, atol=0)
            np.testing.assert_array_equal(rec[:, 1], comp["size"].cpu().numpy().astype(np.float64))
            np.testing.assert_array_equal(rec[:, 3::2].reshape(-1), comp["fold"]["size"].cpu().numpy().astype(np.float64))
        idx = K.flat_index_for(comp)
        probe = torch.from_numpy(np.concatenate([key[:1000], key[:1000] + 1])).cuda()
        got = idx.lookup([probe], [None]).cpu().numpy()
        pk = probe.cpu().numpy()
        p = np.minimum(np.searchsorted(uk, pk), uk.size - 1)
        if idx.ok():
            np.testing.assert_array_equal(got, np.where(uk[p] == pk, p, -1))


def test_merge_counts_sorted_vs_numpy():
    """nvt_count_merge_sorted: (count << 32 | key) rows in source-major, column-minor segments ->
    per column the key-ordered union with summed counts (the owner-side merge of the multi-GPU
    exchange, categorify.py:1054-1070)."""
    from nvtabular_amd import kernels as K

    rng = np.random.default_rng(11)
    G, ncol = 3, 5
    segs, off = [], [0]
    per_col = {j: [] for j in range(ncol)}
    for src in range(G):
        for j in range(ncol):
            m = 0 if (src == 1 and j == 2) else int(rng.integers(1, 40_000))
            if j == 4:
                k = np.unique(rng.integers(-2**31, 2**31 - 1, m, dtype=np.int64)).astype(np.int32)
                if src == 0:
                    k = np.unique(np.concatenate([k, np.array([-2**31, 2**31 - 1], dtype=np.int32)]))
            else:
                k = np.unique(rng.integers(-500 * (j + 1), 30_000, m)).astype(np.int32)
            c = rng.integers(1, 2**20 if j == 3 else 50, k.size).astype(np.int64)
            if j == 3 and k.size:
                c[0] = 2**31 - 1  # the largest count a packed row can carry
            segs.append((c << 32) | (k.astype(np.int64) & 0xFFFFFFFF))
            per_col[j].append((k, c))
            off.append(off[-1] + k.size)
    rows = torch.from_numpy(np.concatenate(segs)).cuda()
    got = K.merge_counts_sorted(rows, off, ncol)
    assert len(got) == ncol
    for j in range(ncol):
        k = np.concatenate([p[0] for p in per_col[j]])
        c = np.concatenate([p[1] for p in per_col[j]])
        uk, inv = np.unique(k, return_inverse=True)
        exp = np.zeros(uk.size, dtype=np.int64)
        np.add.at(exp, inv, c)
        gk, gc = got[j]
        assert gk.dtype == torch.int32 and gc.dtype == torch.int64
        np.testing.assert_array_equal(gk.cpu().numpy(), uk)
        np.testing.assert_array_equal(gc.cpu().numpy(), exp)
    assert K.merge_counts_sorted(rows[:0], [0] * (G * ncol + 1), ncol)[0][0].numel() == 0


def test_exchange_batch_kernels_vs_torch():
    """nvt_exchange_ranges / _hist / _scatter / _unpack (the device work around the collectives of
    dist.merge_counts_many) against the torch formulation they replace."""
    from nvtabular_amd import dist
    from nvtabular_amd import kernels as K

    rng = np.random.default_rng(4)
    dev = "cuda"
    G = 8
    sizes = [200_000, 0, 3, 77_777, 1, 4096, 4097, 50_000, 10, 123_456]
    tabs = []
    for j, m in enumerate(sizes):
        k = np.unique(rng.integers(-2**31, 2**31 - 1, m, dtype=np.int64)).astype(np.int32) if j % 2 == 0 \
            else np.unique(rng.integers(-1000 * j, 5000 * j + 5, m)).astype(np.int32)
        if j == 0:
            k = np.unique(np.concatenate([k, np.array([-2**31, 2**31 - 1], dtype=np.int32)]))
        rng.shuffle(k)  # lists need not be sorted
        c = rng.integers(1, 10_000, k.size).astype(np.int64)
        tabs.append((torch.from_numpy(k).to(dev), torch.from_numpy(c).to(dev)))
    ncol = len(tabs)
    xb = K.ExchangeBatch(tabs)
    got = xb.ranges().cpu().numpy()
    big = np.iinfo(np.int64).max
    for j, (k, c) in enumerate(tabs):
        if k.numel():
            assert got[j].tolist() == [-int(k.min()), int(k.max()), int(c.sum())], j
        else:
            assert got[j].tolist() == [-big, -big, 0]
    los = [int(k.min()) if k.numel() else 0 for k, _ in tabs]
    his = [int(k.max()) if k.numel() else 0 for k, _ in tabs]
    widths = [max(1, -(-(h - l + 1) // G)) for l, h in zip(los, his)]
    owners = [dist._range_owner(k.to(torch.int64), l, h, G) if k.numel() else k.to(torch.int64)
              for (k, _), l, h in zip(tabs, los, his)]
    exp_mat = torch.stack([torch.bincount(o, minlength=G) for o in owners], dim=1)  # [G, ncol]
    mat = xb.hist(los, widths, G)
    assert torch.equal(mat, exp_mat)
    flat = exp_mat.reshape(-1).cpu()
    starts = torch.zeros(G * ncol, dtype=torch.int64)
    starts[1:] = torch.cumsum(flat, 0)[:-1]
    cur = starts.to(dev)
    rows = xb.scatter(los, widths, G, cur)
    assert torch.equal(cur.cpu(), starts + flat)  # cursors advanced to the group ends
    for g in range(G):
        for j, (k, c) in enumerate(tabs):
            a, b = int(starts[g * ncol + j]), int(starts[g * ncol + j] + flat[g * ncol + j])
            sel = owners[j] == g
            exp = ((c[sel] << 32) | (k[sel].to(torch.int64) & 0xFFFFFFFF)).sort().values
            assert torch.equal(rows[a:b].sort().values, exp), (g, j)
    # unpack: segments to arbitrary destinations
    seg_off = [0, 5, 5, 1000, int(rows.numel())]
    dst_off = [int(rows.numel()) - 5 + 8, 0, 8 + int(rows.numel()) - 1000, 8]
    keys, cnts = K.exchange_unpack(rows, seg_off, dst_off, int(rows.numel()) + 16)
    for s in range(4):
        a, b = seg_off[s], seg_off[s + 1]
        w = rows[a:b]
        assert torch.equal(keys[dst_off[s]:dst_off[s] + b - a], ((w << 32) >> 32).to(torch.int32))
        assert torch.equal(cnts[dst_off[s]:dst_off[s] + b - a], w >> 32)


@pytest.mark.parametrize("keyed", [True, False])
def test_clustered_keys_fall_back_to_a_hashed_index(tmp_path, monkeypatch, keyed):
    """Dense ids plus one far outlier: the sort path still aggregates, but the flat index would have
    every key in the same home slot (displacement > 4096) -> without the key directory JoinGroupby /
    TargetEncoding look the groups up through a hashed index built from the sorted keys; with it
    (the default) the crowded bucket is searched by bisection.  Results unchanged either way."""
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K
    from nvtabular_amd import ops

    monkeypatch.setattr(K, "KEYED_IMAGES", keyed)

    rng = np.random.default_rng(17)
    n = 90_000
    k = rng.integers(0, 40_000, n).astype(np.int32)
    k[:5] = 2**31 - 7
    df = pd.DataFrame({"k": k, "x": rng.normal(size=n), "y": (rng.random(n) < 0.4).astype("float32")})
    jg = ops.JoinGroupby(out_path=str(tmp_path / "jg"), stats=["count", "mean"], cont_cols=["x"])
    te = ops.TargetEncoding("y", out_path=str(tmp_path / "te"), kfold=4, fold_seed=1, p_smooth=10)
    wf = nvt.Workflow((["k"] >> jg) + (["k"] >> te)).fit(nvt.Dataset(df))
    assert isinstance(jg._device_stats["k"].index, K.FlatIndex if keyed else K.GroupbyTable)
    got = wf.transform(nvt.Dataset(df)).to_ddf().compute()
    cats = O.join_groupby_fit([df.copy()], ["k"], ["x"], ["count", "mean"], str(tmp_path / "c"))
    exp_j = O.join_groupby_transform(df.copy(), ["k"], cats)
    st, means = O.target_encoding_fit([df.copy()], ["k"], ["y"], str(tmp_path / "c2"), kfold=4, fold_seed=1)
    exp_t = O.target_encoding_transform(df[["k", "y"]].copy(), ["k"], ["y"], st, means, kfold=4, fold_seed=1,
                                        p_smooth=10)
    np.testing.assert_array_equal(got["k_count"].to_numpy(), exp_j["k_count"].to_numpy())
    np.testing.assert_allclose(got["k_x_mean"].to_numpy(), exp_j["k_x_mean"].to_numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(got["TE_k_y"].to_numpy(), exp_t["TE_k_y"].to_numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("kfold,seed", [(5, 42), (3, 7), (8, 1), (2, 0), (16, 123), (10, 2**32 - 1), (100, 5)])
def test_device_folds_are_numpys_mt19937_stream(kfold, seed):
    """nvt_fold_mt19937 == numpy.random.RandomState(seed).choice(arange(kfold), n), bit for bit
    (target_encoding.py:427-439), incl. lengths that end inside a 624-word state."""
    from nvtabular_amd import kernels as K

    for n in (1, 623, 624, 625, 200_001):
        out = torch.empty(n, dtype=torch.uint8, device="cuda")
        K.check(K._lib.load().nvt_fold_mt19937(seed, kfold, n, out.data_ptr(), K.stream_ptr()))
        typ = np.min_scalar_type(kfold * 2)
        exp = np.random.RandomState(seed).choice(np.arange(kfold, dtype=typ), n)
        np.testing.assert_array_equal(out.cpu().numpy(), exp.astype(np.uint8))


@pytest.mark.parametrize("kfold,seed,n", [(5, 42, 3_000_001), (3, 7, 1 << 20), (2, 0, 700_000), (100, 5, 2_500_000),
                                          (8, 1, 9_000_000)])
def test_parallel_folds_are_the_same_mt19937_stream(kfold, seed, n):
    """nvt_fold_mt19937_par (chunks of 2^18 draws reached by jump-ahead polynomials,
    tools/mt_jump_polys.py) == numpy.random.RandomState(seed).choice(arange(kfold), n), bit for
    bit (target_encoding.py:427-439): 5-70 chunks, i.e. jumps through up to 7 polynomials, rejection
    rates 0 .. 37.5 %, a length that ends inside a chunk."""
    import ctypes as C

    from nvtabular_amd import kernels as K

    lib = K._lib.load()
    need = C.c_uint64()
    K.check(lib.nvt_fold_mt19937_par_ws_bytes(n, kfold, C.byref(need)))
    ws = torch.empty(need.value, dtype=torch.uint8, device="cuda")
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    total = torch.zeros(1, dtype=torch.int64, device="cuda")
    K.check(lib.nvt_fold_mt19937_par(seed, kfold, n, out.data_ptr(), ws.data_ptr(), need.value, total.data_ptr(),
                                     K.stream_ptr()))
    assert int(total.item()) >= n
    typ = np.min_scalar_type(kfold * 2)
    exp = np.random.RandomState(seed).choice(np.arange(kfold, dtype=typ), n)
    np.testing.assert_array_equal(out.cpu().numpy(), exp.astype(np.uint8))


def test_fold_column_takes_the_parallel_generator_and_matches_numpy():
    from nvtabular_amd.ops import target_encoding as T

    key = (5, 4242, "cuda:0")
    T._FOLD_CACHE.pop(key, None)
    n = 2_000_000
    col = T._fold_column(n, 5, 4242, torch.device("cuda:0"))
    exp = np.random.RandomState(4242).choice(np.arange(5, dtype=np.uint8), n)
    np.testing.assert_array_equal(col.data.cpu().numpy(), exp)
    # a longer partition later: regenerated (longer prefix of the same stream)
    col2 = T._fold_column(3 * n, 5, 4242, torch.device("cuda:0"))
    np.testing.assert_array_equal(col2.data[:n].cpu().numpy(), exp)


@pytest.mark.parametrize("nparts", [1, 3])
def test_null_keys_are_one_group_on_the_sort_path(tmp_path, nparts):
    """A nullable int32 key column: the rows without a key form one group (the reference groups
    with dropna=False) -- reduced beside the sort path, appended as the last group, looked up
    through FlatIndex.set_null_group.  JoinGroupby statistics and TargetEncoding (with folds)
    against the oracle, over one and three partitions."""
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K
    from nvtabular_amd import ops
    from nvtabular_amd.ops.target_encoding import _FoldDense

    df = _frames(150_000, 6_000, 91)
    rng = np.random.default_rng(3)
    df["k"] = pd.array(df["k"].to_numpy(), dtype="Int32")
    df.loc[rng.random(len(df)) < 0.07, "k"] = pd.NA
    parts = _split(df, nparts)
    stats = ["count", "sum", "mean", "min", "max"]
    jg = ops.JoinGroupby(out_path=str(tmp_path / "jg"), stats=stats, cont_cols=["x", "y"])
    te = ops.TargetEncoding(["y"], out_path=str(tmp_path / "te"), kfold=5, fold_seed=42, p_smooth=20)
    wf = nvt.Workflow((["k"] >> jg) + (["k"] >> te)).fit(nvt.Dataset(parts))
    assert isinstance(jg._device_stats["k"].index, K.FlatIndex)
    assert isinstance(te._device_stats["__fold___k"], _FoldDense)
    got = wf.transform(nvt.Dataset(parts)).to_ddf().compute()
    oparts = [p.copy() for p in parts]
    for p in oparts:   # what pandas' parquet reader hands the reference: float64 with NaN
        p["k"] = p["k"].astype("float64")
    cats = O.join_groupby_fit([p.copy() for p in oparts], ["k"], ["x", "y"], stats, str(tmp_path / "c"))
    exp = O.join_groupby_transform(pd.concat(oparts, ignore_index=True), ["k"], cats)
    for c in exp.columns:
        if c.endswith("_count"):
            np.testing.assert_array_equal(got[c].to_numpy(), exp[c].to_numpy(), err_msg=c)
        else:
            np.testing.assert_allclose(got[c].to_numpy().astype("float64"), exp[c].to_numpy().astype("float64"),
                                       rtol=2e-5, atol=1e-6, err_msg=c)
    tstats, means = O.target_encoding_fit([p.copy() for p in oparts], ["k"], ["y"], str(tmp_path / "ct"),
                                          kfold=5, fold_seed=42)
    texp = pd.concat([O.target_encoding_transform(p[["k", "y"]].copy(), ["k"], ["y"], tstats, means, kfold=5,
                                                  fold_seed=42, p_smooth=20) for p in oparts], ignore_index=True)
    np.testing.assert_allclose(got["TE_k_y"].to_numpy(), texp["TE_k_y"].to_numpy(), rtol=1e-5, atol=1e-6)
    isnull = pd.concat(parts, ignore_index=True)["k"].isna().to_numpy()
    assert isnull.any() and np.isfinite(got["k_x_sum"].to_numpy()[isnull]).all()   # the null group has statistics
    assert (got["k_count"].to_numpy()[isnull] == 0).all()   # ... and the count of its (null) keys is 0
