"""Parity of the HIP path (through the C ABI) against the CPU oracle.

Integer / index results must be bit-exact; floating-point statistics within the
tolerance written at each assert (north star: 1e-6 relative for means / stds).
"""
import numpy as np
import pandas as pd
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _dev():
    return torch.device("cuda", 0)


def _nullable_int_frame(rng, n, card, null_frac, dtype="int32", zipf=1.2):
    """pandas frame with an Arrow-style nullable int column + the float64 view
    pandas' own reader would have produced (what the oracle consumes)."""
    raw = np.minimum(rng.zipf(zipf, n), card).astype(np.int64)
    ids = (raw * 2654435761) % (2**31 - 1) - (2**30)  # spread, some negative
    ids = ids.astype(dtype)
    mask = rng.random(n) < null_frac
    return ids, mask


def _device_col(ids, mask):
    from nvtabular_amd.device import DeviceColumn, pack_bitmap

    data = torch.from_numpy(ids).to(_dev())
    valid = torch.from_numpy(pack_bitmap(~mask)).to(_dev()) if mask.any() else None
    return DeviceColumn(data, valid)


def _oracle_series(ids, mask):
    s = pd.Series(ids.astype("float64") if mask.any() else ids)
    if mask.any():
        s[mask] = np.nan
    return s


@pytest.mark.parametrize("dtype", ["int32", "int64"])
@pytest.mark.parametrize("null_frac", [0.0, 0.2])
@pytest.mark.parametrize("n,card", [(1, 1), (7, 3), (1000, 50), (200_003, 5000), (300_000, 10**9)])
def test_categorify_fit_transform_vs_oracle(tmp_path, dtype, null_frac, n, card):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops
    from nvtabular_amd.device import DeviceFrame

    rng = np.random.default_rng(n + card)
    ids, mask = _nullable_int_frame(rng, n, card, null_frac, dtype)
    if n > 10:
        ids[3] = np.iinfo(dtype).min  # the hash tables' empty-slot sentinel is a legal key
        mask[3] = False
    frame = DeviceFrame({"c": _device_col(ids, mask)})
    odf = pd.DataFrame({"c": _oracle_series(ids, mask)})

    cats = ["c"] >> ops.Categorify(out_path=str(tmp_path / "gpu"))
    wf = nvt.Workflow(cats)
    wf.fit(nvt.Dataset(frame))
    got = wf.transform(frame)["c"].data.cpu().numpy()

    paths = O.categorify_fit([odf], ["c"], str(tmp_path / "cpu"), tie_break="stable")
    exp = O.categorify_transform(odf, ["c"], paths)["c"].to_numpy()
    assert got.dtype == np.int64
    np.testing.assert_array_equal(got, exp)

    # on-disk artefacts (Appendix B of SURVEY): same vocabulary, sizes and labels
    gv = pd.read_parquet(tmp_path / "gpu" / "categories" / "unique.c.parquet")
    ov = pd.read_parquet(paths["c"])
    np.testing.assert_array_equal(gv.index.to_numpy(), ov.index.to_numpy())
    np.testing.assert_array_equal(gv["c"].to_numpy().astype("int64"), ov["c"].to_numpy().astype("int64"))
    np.testing.assert_array_equal(gv["c_size"].to_numpy(), ov["c_size"].to_numpy())
    gm = pd.read_parquet(tmp_path / "gpu" / "categories" / "meta.c.parquet")
    om = pd.read_parquet(tmp_path / "cpu" / "categories" / "meta.c.parquet")
    assert gm["num_observed"].tolist() == [int(v) for v in om["num_observed"].tolist()]
    assert gm["num_indices"].tolist() == om["num_indices"].tolist()

    # reference tie order (unstable numpy sort) agrees after canonicalising ties
    pv = pd.read_parquet(O.categorify_fit([odf], ["c"], str(tmp_path / "ref"), tie_break="pandas")["c"])
    assert pv["c_size"].tolist() == gv["c_size"].tolist()
    for size, blk in pv.groupby("c_size"):
        assert set(blk["c"].astype("int64")) == set(gv["c"][gv["c_size"] == size].astype("int64"))


@pytest.mark.parametrize("kw", [
    dict(freq_threshold=3),
    dict(max_size=40),
    dict(freq_threshold=2, num_buckets=7),
    dict(max_size=30, num_buckets=5),
    dict(dtype=np.int32),
])
def test_categorify_options_vs_oracle(tmp_path, kw):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops
    from nvtabular_amd.device import DeviceFrame

    rng = np.random.default_rng(11)
    ids, mask = _nullable_int_frame(rng, 5000, 300, 0.05)
    frame = DeviceFrame({"c": _device_col(ids, mask)})
    odf = pd.DataFrame({"c": _oracle_series(ids, mask)})
    with _nowarn():
        op = ops.Categorify(out_path=str(tmp_path / "gpu"), **kw)
    wf = nvt.Workflow(["c"] >> op).fit(nvt.Dataset(frame))
    got = wf.transform(frame)["c"].data.cpu().numpy()
    okw = {k: v for k, v in kw.items() if k != "dtype"}
    paths = O.categorify_fit([odf], ["c"], str(tmp_path / "cpu"), tie_break="stable", **okw)
    exp = O.categorify_transform(odf, ["c"], paths, num_buckets=kw.get("num_buckets"),
                                 dtype=kw.get("dtype"))["c"].to_numpy()
    assert got.dtype == (np.dtype(kw["dtype"]) if "dtype" in kw else np.int64)
    np.testing.assert_array_equal(got, exp)
    gm = pd.read_parquet(tmp_path / "gpu" / "categories" / "meta.c.parquet")
    assert gm["num_observed"].sum() == len(odf)  # test_categorify.py:383-388


class _nowarn:
    def __enter__(self):
        import warnings

        self._c = warnings.catch_warnings()
        self._c.__enter__()
        warnings.simplefilter("ignore")

    def __exit__(self, *a):
        return self._c.__exit__(*a)


def test_categorify_multi_partition_merge(tmp_path):
    """Tree merge across partitions == one-shot fit (test_categorify.py:668-704 spirit)."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(5)
    df = pd.DataFrame({
        "a": rng.integers(0, 2000, 40_000).astype("int64"),
        "b": rng.integers(-50, 50, 40_000).astype("int32"),
    })
    parts = [df.iloc[i : i + 7000].reset_index(drop=True) for i in range(0, len(df), 7000)]
    wf = nvt.Workflow(["a", "b"] >> ops.Categorify(out_path=str(tmp_path / "g")))
    wf.fit(nvt.Dataset(parts))
    got = wf.transform(nvt.Dataset(parts)).to_ddf().compute()
    paths = O.categorify_fit(parts, ["a", "b"], str(tmp_path / "c"), tie_break="stable")
    exp = O.categorify_transform(df, ["a", "b"], paths)
    np.testing.assert_array_equal(got["a"].to_numpy(), exp["a"].to_numpy())
    np.testing.assert_array_equal(got["b"].to_numpy(), exp["b"].to_numpy())


# ---- golden vectors of the reference, through the Workflow API (strings) ----
def test_golden_categorify_lists(tmp_path):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    for freq in (0, 1, 2):
        df = pd.DataFrame({
            "Authors": [["User_A"], ["User_A", "User_E"], ["User_B", "User_C"], ["User_C"]],
            "Engaging User": ["User_B", "User_B", "User_A", "User_D"],
            "Post": [1, 2, 3, 4],
        })
        cats = ["Authors", "Engaging User"] >> ops.Categorify(out_path=str(tmp_path / str(freq)),
                                                              freq_threshold=freq)
        out = nvt.Workflow(cats + ["Post"]).fit_transform(nvt.Dataset(df)).to_ddf().compute()
        compare = [list(r) for r in out["Authors"].tolist()]
        # tests/unit/ops/test_categorify.py:154-157
        assert compare == ([[3], [3, 6], [5, 4], [4]] if freq < 2 else [[3], [3, 2], [2, 4], [4]])
        assert out["Post"].tolist() == [1, 2, 3, 4]


@pytest.mark.parametrize("grouped", [True, False])
@pytest.mark.parametrize("kind", ["joint", "combo"])
def test_golden_categorify_multi(tmp_path, grouped, kind):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    df = pd.DataFrame({
        "Author": ["User_A", "User_E", "User_B", "User_C"],
        "Engaging User": ["User_B", "User_B", "User_A", "User_D"],
        "Post": [1, 2, 3, 4],
    })
    names = [["Author", "Engaging User"]] if grouped else ["Author", "Engaging User"]
    cats = names >> ops.Categorify(out_path=str(tmp_path), encode_type=kind)
    out = nvt.Workflow(cats + ["Post"]).fit_transform(nvt.Dataset(df)).to_ddf().compute()
    # tests/unit/ops/test_categorify.py:181-216
    if grouped and kind == "joint":
        assert out["Author"].tolist() == [4, 7, 3, 5]
        assert out["Engaging User"].tolist() == [3, 3, 4, 6]
    elif grouped:
        assert out["Author_Engaging User"].tolist() == [3, 6, 4, 5]
    else:
        assert out["Author"].tolist() == [3, 6, 4, 5]
        assert out["Engaging User"].tolist() == [3, 3, 4, 5]


def test_golden_categorify_combo_with_null(tmp_path):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    # tests/unit/ops/test_categorify.py:290-299
    df = pd.DataFrame({
        "Author": [np.nan, "User_E", "User_B", "User_A"],
        "Engaging User": ["User_C", "User_B", "User_A", "User_D"],
        "Post": [1, 2, 3, 4],
    })
    names = [["Author", "Engaging User"], ["Author"], ["Engaging User"]]
    cats = names >> ops.Categorify(out_path=str(tmp_path), encode_type="combo")
    out = nvt.Workflow(cats + ["Post"]).fit_transform(nvt.Dataset(df)).to_ddf().compute()
    assert out["Author"].tolist() == [1, 5, 4, 3]
    assert out["Engaging User"].tolist() == [5, 4, 3, 6]
    assert out["Author_Engaging User"].tolist() == [3, 6, 5, 4]


def test_golden_na_value_count(tmp_path):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    # tests/unit/ops/test_categorify.py:99-121
    df = pd.DataFrame({
        "productID": ["B00406YHLI"] * 5 + ["B002YXS8E6"] * 5 + ["B00011KM38"] * 2 + [np.nan] * 3,
        "brand": ["Coby"] * 5 + [np.nan] * 5 + ["Cooler Master"] * 2 + ["Asus"] * 3,
    })
    wf = nvt.Workflow(["brand", "productID"] >> ops.Categorify(out_path=str(tmp_path)))
    wf.fit(nvt.Dataset(df))
    wf.transform(nvt.Dataset(df)).to_ddf().compute()
    m1 = pd.read_parquet(tmp_path / "categories" / "meta.brand.parquet")
    m2 = pd.read_parquet(tmp_path / "categories" / "meta.productID.parquet")
    assert m1["kind"].iloc[1] == "null" and m1["num_observed"].iloc[1] == 5
    assert m2["num_observed"].iloc[1] == 3


# ---- continuous path ---------------------------------------------------------
@pytest.mark.parametrize("dtype", ["float32", "float64", "int32", "int64"])
def test_fill_normalize_vs_oracle(dtype):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops
    from nvtabular_amd.device import DeviceFrame

    rng = np.random.default_rng(3)
    n = 100_003
    x = np.floor(rng.lognormal(2, 2, n)).astype(dtype)
    mask = rng.random(n) < 0.3
    if dtype.startswith("float"):
        xs = x.copy()
        xs[mask] = np.nan
        col = _device_col(xs, np.zeros(n, bool))
        odf = pd.DataFrame({"x": xs})
    else:
        col = _device_col(x, mask)
        odf = pd.DataFrame({"x": _oracle_series(x, mask)})
    frame = DeviceFrame({"x": col})
    parts = [odf.iloc[:40_000].copy(), odf.iloc[40_000:].copy()]

    conts = ["x"] >> ops.FillMissing(fill_val=1) >> ops.Normalize()
    wf = nvt.Workflow(conts).fit(nvt.Dataset(frame))
    op = wf.output_node.op
    filled = [O.fill_missing(p.copy(), ["x"], 1) for p in parts]
    mom = O.custom_moments(filled, ["x"])
    # north star: means / stds within 1e-6 relative
    assert abs(op.means["x"] - mom["mean"]["x"]) <= 1e-6 * abs(mom["mean"]["x"])
    assert abs(op.stds["x"] - mom["std"]["x"]) <= 1e-6 * abs(mom["std"]["x"])
    got = wf.transform(frame)["x"].data.cpu().numpy()
    assert got.dtype == np.float64
    ofull = O.fill_missing(odf.copy(), ["x"], 1)
    exp = O.normalize_transform(ofull, ["x"], {"x": op.means["x"]}, {"x": op.stds["x"]})["x"]
    tol = 1e-6 if dtype == "float32" else 1e-12  # pandas keeps float32 arithmetic for float32 input
    np.testing.assert_allclose(got, exp.to_numpy(), rtol=tol, atol=tol)


def test_normalize_nulls_stay_nan_and_std_zero():
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    df = pd.DataFrame({"a": [7.0] * 10, "b": [1.0, np.nan, 3.0, 4.0, np.nan, 6.0, 7.0, 8.0, 9.0, 10.0]})
    wf = nvt.Workflow(["a", "b"] >> ops.Normalize()).fit(nvt.Dataset(df))
    out = wf.transform(df)
    assert (out["a"] == 0).all()  # tests/unit/ops/test_normalize.py:110-117
    assert np.isnan(out["b"][1]) and np.isnan(out["b"][4])
    exp = (df["b"] - df["b"].mean()) / df["b"].std()
    np.testing.assert_allclose(out["b"].to_numpy(), exp.to_numpy(), rtol=1e-12, equal_nan=True)


@pytest.mark.parametrize("add_binary_cols", [True, False])
def test_golden_fill_missing(add_binary_cols):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    # tests/unit/ops/test_fill.py:61-85
    df = pd.DataFrame({"x": [1.0, np.nan, 3.0], "y": [np.nan, 2.0, np.nan]})
    wf = nvt.Workflow(["x", "y"] >> ops.FillMissing(42, add_binary_cols=add_binary_cols))
    out = wf.transform(df)
    assert out["x"].tolist() == [1.0, 42.0, 3.0] and out["y"].tolist() == [42.0, 2.0, 42.0]
    if add_binary_cols:
        assert out["x_filled"].tolist() == [False, True, False]
        assert out["y_filled"].dtype == bool


def test_hash_bucket_vs_oracle():
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(0)
    df = pd.DataFrame({
        "a": rng.integers(-(2**62), 2**62, 50_001),
        "b": rng.integers(-(2**31), 2**31 - 1, 50_001).astype("int32"),
    })
    wf = nvt.Workflow(["a", "b"] >> ops.HashBucket({"a": 100, "b": 2**20}))
    out = wf.transform(df)
    exp = O.hash_bucket_op(df.copy(), {"a": 100, "b": 2**20})
    assert out["a"].dtype == np.int32
    np.testing.assert_array_equal(out["a"].to_numpy(), exp["a"].to_numpy())
    np.testing.assert_array_equal(out["b"].to_numpy(), exp["b"].to_numpy())


# ---- JoinGroupby / TargetEncoding ------------------------------------------------
@pytest.mark.parametrize("multi", [True, False])
def test_golden_joingroupby(tmp_path, multi):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    # tests/unit/ops/test_join.py:62-92
    df = pd.DataFrame({
        "Author": ["User_A", "User_A", "User_A", "User_B"],
        "Engaging-User": ["User_B", "User_B", "User_C", "User_C"],
        "Cost": [100.0, 200.0, 300.0, 400.0],
        "Post": [1, 2, 3, 4],
    })
    groups = [["Author", "Engaging-User"]] if multi else ["Author"]
    feats = groups >> ops.JoinGroupby(out_path=str(tmp_path), stats=["sum"], cont_cols=["Cost"])
    out = nvt.Workflow(feats + "Post").fit_transform(nvt.Dataset(df)).to_ddf().compute()
    if multi:
        assert out["Author_Engaging-User_Cost_sum"].tolist() == [300.0, 300.0, 300.0, 400.0]
    else:
        assert out["Author_Cost_sum"].tolist() == [600.0, 600.0, 600.0, 400.0]


def test_golden_joingroupby_dependency(tmp_path):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    # tests/unit/ops/test_join.py:32-58
    df = pd.DataFrame({
        "Author": ["User_A", "User_A", "User_A", "User_B", "User_B"],
        "Cost": [100.0, 200.0, 300.0, 400.0, 400.0],
    })
    norm = ["Cost"] >> ops.NormalizeMinMax() >> ops.Rename(postfix="_normalized")
    feats = ["Author"] >> ops.JoinGroupby(out_path=str(tmp_path), stats=["sum"], cont_cols=norm)
    out = nvt.Workflow(feats).fit_transform(nvt.Dataset(df)).to_ddf().compute()
    assert out["Author_Cost_normalized_sum"].tolist() == [1.0, 1.0, 1.0, 2.0, 2.0]


def test_joingroupby_stats_vs_oracle(tmp_path):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(3)
    n = 30_000
    df = pd.DataFrame({
        "k": rng.integers(0, 500, n),
        "j": rng.integers(0, 7, n).astype("int32"),
        "x": rng.normal(size=n),
        "y": rng.normal(size=n).astype("float32"),
    })
    df.loc[rng.random(n) < 0.05, "x"] = np.nan
    parts = [df.iloc[:9000].reset_index(drop=True), df.iloc[9000:].reset_index(drop=True)]
    stats = ["count", "sum", "mean", "std", "var", "min", "max"]
    groups = ["k", ["k", "j"]]
    feats = groups >> ops.JoinGroupby(out_path=str(tmp_path / "g"), stats=stats, cont_cols=["x", "y"])
    wf = nvt.Workflow(feats).fit(nvt.Dataset(parts))
    got = wf.transform(nvt.Dataset(parts)).to_ddf().compute()
    cats = O.join_groupby_fit([p.copy() for p in parts], groups, ["x", "y"], stats, str(tmp_path / "c"))
    exp = O.join_groupby_transform(df.copy(), groups, cats)
    # column order: the executor returns columns in output-schema (column_mapping) order
    assert sorted(got.columns) == sorted(exp.columns)
    for c in exp.columns:
        if c.endswith("_count"):
            assert got[c].dtype == np.int32
            np.testing.assert_array_equal(got[c].to_numpy(), exp[c].to_numpy())
        else:
            # fp64 atomics add in a different order than pandas: 1e-6 relative
            np.testing.assert_allclose(got[c].to_numpy().astype("float64"),
                                       exp[c].to_numpy().astype("float64"), rtol=2e-5, atol=1e-6,
                                       err_msg=c)
            if c.rsplit("_", 1)[1] in ("mean", "std", "var"):
                assert got[c].dtype == np.float32


@pytest.mark.parametrize("kfold", [1, 3])
@pytest.mark.parametrize("fold_seed", [None, 42])
def test_target_encoding_vs_oracle(tmp_path, kfold, fold_seed):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(9)
    n = 20_000
    df = pd.DataFrame({
        "a": rng.integers(0, 300, n),
        "b": rng.integers(0, 11, n).astype("int32"),
        "y": (rng.random(n) < 0.3).astype("float32"),
        "z": rng.normal(size=n),
    })
    parts = [df.iloc[:8000].reset_index(drop=True), df.iloc[8000:].reset_index(drop=True)]
    groups = ["a", ["a", "b"]]
    te = groups >> ops.TargetEncoding(["y", "z"], out_path=str(tmp_path / "g"), kfold=kfold,
                                      fold_seed=fold_seed, p_smooth=20)
    wf = nvt.Workflow(te).fit(nvt.Dataset(parts))
    got = wf.transform(nvt.Dataset(parts)).to_ddf().compute()
    oparts = [p.copy() for p in parts]
    stats, means = O.target_encoding_fit(oparts, groups, ["y", "z"], str(tmp_path / "c"),
                                         kfold=kfold, fold_seed=fold_seed)
    exp = pd.concat([
        O.target_encoding_transform(p[["a", "b", "y", "z"]].copy(), groups, ["y", "z"], stats, means,
                                    kfold=kfold, fold_seed=fold_seed, p_smooth=20)
        for p in parts
    ], ignore_index=True)
    assert list(got.columns) == list(exp.columns)
    for c in exp.columns:
        assert got[c].dtype == np.float32
        np.testing.assert_allclose(got[c].to_numpy(), exp[c].to_numpy(), rtol=1e-5, atol=1e-6,
                                   err_msg=c)


# ---- C-ABI level properties at larger sizes --------------------------------------
def test_vocab_sort_is_sorted_and_a_permutation():
    from nvtabular_amd import kernels as K

    g = torch.Generator(device="cuda").manual_seed(1)
    n = 3_000_017
    keys = torch.randint(-(2**31), 2**31 - 1, (n,), device="cuda", dtype=torch.int64, generator=g)
    keys = torch.unique(keys).to(torch.int32)
    n = keys.numel()
    counts = torch.randint(1, 50, (n,), device="cuda", dtype=torch.int64, generator=g)
    counts[:5] = torch.tensor([2**33, 1, 2**20, 7, 2**33], device="cuda")
    k0, c0 = keys.clone(), counts.clone()
    K.vocab_sort(keys, counts)
    torch.cuda.synchronize()
    dc = counts[1:] - counts[:-1]
    assert bool((dc <= 0).all())
    ties = dc == 0
    assert bool((keys[1:][ties] > keys[:-1][ties]).all())
    # same multiset of pairs
    o0 = torch.argsort(k0)
    o1 = torch.argsort(keys)
    assert torch.equal(k0[o0], keys[o1]) and torch.equal(c0[o0], counts[o1])


def test_count_roundtrip_large():
    """Size-independent check at bench-like scale: sum of counts == rows, labels of the
    fitted data never hit OOV, label -> key lookup inverts the encode."""
    from nvtabular_amd import kernels as K

    g = torch.Generator(device="cuda").manual_seed(2)
    n = 8_000_000
    u = torch.rand(n, device="cuda", generator=g)
    keys = ((1.0 / (1.0 - u * 0.999999)) ** 1.3).to(torch.int64)  # heavy tail
    keys = (keys * 2654435761 % (2**31 - 1)).to(torch.int32)
    tab, st = K.count_into_new_table([keys], [None], hint=1 << 12)
    vk, vc = tab.compact()
    assert int(vc.sum().item()) == n
    assert vk.numel() == torch.unique(keys).numel()
    K.vocab_sort(vk, vc)
    enc = K.EncodeTable(vk, 3)
    labels = enc.encode(keys, None, 1, 2)
    assert int(labels.min().item()) == 3 and int(labels.max().item()) == 2 + vk.numel()
    assert torch.equal(vk[(labels - 3)], keys)


# ---- atomic-free dense counting (nvt_dense_count_*): every path, exact ----------------
@pytest.mark.parametrize("dtype", ["int32", "int64"])
@pytest.mark.parametrize("card,n", [(5, 1000), (3000, 200_000), (50_000, 400_000),
                                    (10**9, 1_500_000)])
@pytest.mark.parametrize("weighted", [False, True])
def test_dense_count_paths_vs_numpy(dtype, card, n, weighted):
    from nvtabular_amd import kernels as K
    from nvtabular_amd.device import pack_bitmap

    rng = np.random.default_rng(card % 1000 + n)
    ids, mask = _nullable_int_frame(rng, n, card, 0.1, dtype)
    ids[5] = np.iinfo(dtype).min
    mask[5] = False
    w = rng.integers(1, 1000, n).astype("int64") if weighted else None
    keys = torch.from_numpy(ids).cuda()
    valid = torch.from_numpy(pack_bitmap(~mask)).cuda()
    wt = torch.from_numpy(w).cuda() if weighted else None
    ww = w if weighted else np.ones(n, dtype="int64")
    exp = pd.Series(ww[~mask]).groupby(ids[~mask]).sum()
    distinct = len(exp)
    paths = [(p, False) for p in K.PATH_ORDER]
    if dtype == "int32" and not weighted:
        paths += [(p, True) for p in (1, 2, 3)]  # hot-key filter variants
    for path, hot in paths:
        job = K.DenseCountJob(keys, valid, wt, hint=distinct)
        job.path = path  # force every kernel path (the driver escalates along PATH_ORDER on overflow)
        job.hot = hot
        k, c, nulls, info = K.dense_count_many([job])[0]
        assert info["path"] == path or path in (6, 0)  # LDS-table paths may escalate
        got = pd.Series(c.cpu().numpy(), index=k.cpu().numpy()).sort_index()
        assert got.index.is_unique
        np.testing.assert_array_equal(got.index.to_numpy(), exp.index.to_numpy())
        np.testing.assert_array_equal(got.to_numpy(), exp.to_numpy())
        assert nulls == int(ww[mask].sum())
        assert info["max_count"] >= int(exp.max()) and info["rows"] == n


@pytest.mark.parametrize("dtype", ["int32", "int64"])
@pytest.mark.parametrize("n", [2, 3, 100, 4097, 8192, 8193, 50_000, 300_000])
def test_vocab_sort_vs_numpy(dtype, n):
    """(count desc, key asc) on both code paths: LDS bitonic (n <= 8192) and LSD radix."""
    from nvtabular_amd import kernels as K

    rng = np.random.default_rng(n)
    info = np.iinfo(dtype)
    keys = rng.choice(np.arange(-3 * n, 3 * n, dtype=np.int64), size=n, replace=False).astype(dtype)
    keys[0], keys[1 % n] = info.min, info.max
    keys = np.unique(keys)
    m = len(keys)
    rng.shuffle(keys)
    counts = rng.integers(1, 6, m).astype("int64")  # lots of ties
    counts[: min(3, m)] = [2**40, 1, 2**33][: min(3, m)]
    order = np.lexsort((keys, -counts))
    tk, tc = torch.from_numpy(keys).cuda(), torch.from_numpy(counts).cuda()
    K.vocab_sort(tk, tc)
    np.testing.assert_array_equal(tk.cpu().numpy(), keys[order])
    np.testing.assert_array_equal(tc.cpu().numpy(), counts[order])


def test_virtual_shard_vocab_merge_matches_single_shot():
    """the multi-GPU vocabulary merge with G virtual shards on ONE GPU.
    Same device steps as nvtabular_amd.dist.merge_counts (owner = h32(key) % G via the HIP
    hash kernel, owner-side weighted dense count), with the all-to-all / all-gather replaced
    by in-process routing -- the RCCL choreography itself is covered by test_dist_gloo.py."""
    from nvtabular_amd import dist, kernels as K

    G = 4
    rng = np.random.default_rng(42)
    n = 400_000
    ids, _ = _nullable_int_frame(rng, n, 50_000, 0.0, "int32")
    shards = np.array_split(ids, G)
    per_rank = []
    for part in shards:
        k, c, _, _ = K.dense_count(torch.from_numpy(part).cuda(), None, None, hint=0)
        per_rank.append((k, c))
    # route every (key, count) row to its owner, merge there, gather
    inbox = [[] for _ in range(G)]
    for k, c in per_rank:
        owner = dist._hip_owner([k], G).to(torch.int64)
        assert int(owner.min()) >= 0 and int(owner.max()) < G
        for g in range(G):
            m = owner == g
            inbox[g].append((k[m], c[m]))
    merged = []
    for g in range(G):
        mk, mc = dist._hip_merge_counts(torch.cat([x[0] for x in inbox[g]]),
                                        torch.cat([x[1] for x in inbox[g]]))
        merged.append((mk, mc))
    gk = torch.cat([m[0] for m in merged]).cpu().numpy()
    gc = torch.cat([m[1] for m in merged]).cpu().numpy()
    got = pd.Series(gc, index=gk).sort_index()
    exp = pd.Series(1, index=ids).groupby(level=0).sum().sort_index()
    assert got.index.is_unique  # a key has exactly one owner
    np.testing.assert_array_equal(got.index.to_numpy(), exp.index.to_numpy())
    np.testing.assert_array_equal(got.to_numpy(), exp.to_numpy())
    # owners agree with the oracle's hash definition
    import oracle as O

    k0 = per_rank[0][0]
    np.testing.assert_array_equal(dist._hip_owner([k0], G).cpu().numpy(),
                                  (O.nvt_hash32(k0.cpu().numpy()) % G).astype("int32"))


@pytest.mark.parametrize("dtype", ["float32", "float64", "int32"])
def test_fill_clip_log_vs_oracle(dtype):
    """The reference benchmark's default continuous branch: FillMissing >> Clip(min_value=0) >>
    LogOp (dask-nvtabular-criteo-benchmark.py:201-204; clip.py:49-55, logop.py:43-53)."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(4)
    n = 50_003
    x = (rng.normal(size=n) * 50).astype(dtype)
    df = pd.DataFrame({"x": x.astype("float64") if dtype == "int32" else x})
    df.loc[rng.random(n) < 0.2, "x"] = np.nan
    gdf = df.copy()
    if dtype == "int32":
        gdf["x"] = pd.array(np.where(df["x"].isna(), 0, df["x"]).astype("int32"), dtype="Int32")
        gdf.loc[df["x"].isna(), "x"] = pd.NA
    wf = nvt.Workflow(["x"] >> ops.FillMissing() >> ops.Clip(min_value=0) >> ops.LogOp())
    got = wf.transform(gdf)["x"].to_numpy()
    assert got.dtype == np.float32
    ref = O.fill_missing(df.copy(), ["x"], 0)
    ref = O.clip_transform(ref, ["x"], min_value=0)
    ref = O.logop_transform(ref, ["x"])["x"].to_numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-7)  # float32 log: last-ulp differences
    # Clip alone keeps dtype and nulls (clip.py:49-55)
    c = nvt.Workflow(["x"] >> ops.Clip(min_value=-10, max_value=25)).transform(gdf)["x"]
    ec = O.clip_transform(df, ["x"], -10, 25)["x"]
    np.testing.assert_array_equal(np.isnan(c.to_numpy(dtype="float64")), ec.isna().to_numpy())
    np.testing.assert_allclose(c.to_numpy(dtype="float64"), ec.to_numpy(), rtol=0, atol=0, equal_nan=True)
    with pytest.raises(ValueError):
        ops.Clip()


@pytest.mark.parametrize("card", [40, 30_000, 10**9])
def test_dense_count_cold_start_presample(card):
    """No cardinality hint + a long column: the first path comes from the distinct count of a
    256 K-row prefix (kernels._presample); the result must be exact whatever it picks."""
    from nvtabular_amd import kernels as K
    from nvtabular_amd.device import pack_bitmap

    n = K.SAMPLE_MIN_ROWS + 4099
    rng = np.random.default_rng(card % 97)
    ids, mask = _nullable_int_frame(rng, n, card, 0.05, "int32", zipf=1.3)
    keys = torch.from_numpy(ids).cuda()
    valid = torch.from_numpy(pack_bitmap(~mask)).cuda()
    job = K.DenseCountJob(keys, valid, None, hint=0)
    k, c, nulls, info = K.dense_count_many([job])[0]
    exp = pd.Series(np.ones(int((~mask).sum()), dtype="int64")).groupby(ids[~mask]).sum()
    got = pd.Series(c.cpu().numpy(), index=k.cpu().numpy()).sort_index()
    np.testing.assert_array_equal(got.index.to_numpy(), exp.index.to_numpy())
    np.testing.assert_array_equal(got.to_numpy(), exp.to_numpy())
    assert nulls == int(mask.sum()) and info["rows"] == n
    if card <= 40:
        assert info["path"] in (6, 0)
    if card >= 30_000 and len(exp) > K.PATH_S_MAX_DISTINCT:
        assert info["path"] not in (6, 0)
    assert job.hint > 0  # the estimate replaced the missing hint


def test_hashed_cross_and_bucketize_vs_oracle():
    """hashed_cross.py:56-67, bucketize.py:76-94; reference tests: tests/unit/ops/test_ops.py:189-236
    (range + determinism for the cross, range for the buckets) -- here bit-exact vs the oracle."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(21)
    n = 40_001
    df = pd.DataFrame({"a": rng.integers(-5, 1000, n).astype("int32"), "b": rng.integers(0, 10**12, n),
                       "s": rng.choice(["u", "v", "w", "xyz"], n),
                       "x": rng.normal(size=n) * 3, "y": rng.integers(-10, 200, n).astype("int32")})
    df.loc[rng.random(n) < 0.1, "x"] = np.nan
    cross = [["a", "b", "s"]] >> ops.HashedCross(10)
    got = nvt.Workflow(cross).fit_transform(nvt.Dataset(df)).to_ddf().compute()
    exp = O.hashed_cross(df, ["a", "b", "s"], 10)
    assert got.columns.tolist() == ["a_X_b_X_s"] and got["a_X_b_X_s"].dtype == np.int32
    np.testing.assert_array_equal(got["a_X_b_X_s"].to_numpy(), exp["a_X_b_X_s"].to_numpy())
    assert got["a_X_b_X_s"].between(0, 9).all()

    bounds = {"x": [-1, 0, 1], "y": [-4, 100]}
    bk = ["x", "y"] >> ops.Bucketize(bounds)
    wf = nvt.Workflow(bk)
    g2 = wf.fit_transform(nvt.Dataset(df)).to_ddf().compute()
    e2 = O.bucketize(df, bounds)
    for c in ("x", "y"):
        assert g2[c].dtype == np.int32
        np.testing.assert_array_equal(g2[c].to_numpy(), e2[c].to_numpy())
    assert wf.output_schema["x"].tags == (nvt.Tags.CATEGORICAL,) or nvt.Tags.CATEGORICAL in wf.output_schema["x"].tags
    with pytest.raises(TypeError):
        ops.Bucketize(3)


@pytest.mark.parametrize("path", [1, 2, 3])
@pytest.mark.parametrize("shape", ["powerlaw", "uniform", "sorted", "one_key"])
def test_hot_filter_counts_exact_for_any_hot_set(path, shape):
    """NVT_PATH_HOT: the rows of the sampled hot keys are counted in LDS, the others are
    partitioned -- exact whatever the sample saw: power law (most rows hot), uniform (the
    sample finds nothing worth keeping: filter switched off), sorted input (the sample's hot
    keys are local), a single key (everything hot, empty partition)."""
    from nvtabular_amd import kernels as K
    from nvtabular_amd.device import pack_bitmap

    rng = np.random.default_rng(path * 7 + len(shape))
    n = 3_000_001
    if shape == "powerlaw":
        ids = (rng.zipf(1.15, n) % 2_000_000).astype("int32") * 3 - 1000
    elif shape == "uniform":
        ids = rng.integers(-2**31, 2**31 - 1, n).astype("int32")
    elif shape == "sorted":
        ids = np.sort((rng.zipf(1.3, n) % 500_000).astype("int32"))
    else:
        ids = np.full(n, 77, dtype="int32")
    ids[123] = np.iinfo("int32").min  # the empty-slot sentinel as a key
    mask = rng.random(n) < 0.07
    keys = torch.from_numpy(ids).cuda()
    valid = torch.from_numpy(pack_bitmap(~mask)).cuda()
    exp = pd.Series(np.ones(n, dtype="int64")[~mask]).groupby(ids[~mask]).sum()
    for v in (valid, None):
        e = exp if v is not None else pd.Series(np.ones(n, dtype="int64")).groupby(ids).sum()
        job = K.DenseCountJob(keys, v, None, hint=len(e))
        job.path, job.hot = path, True
        k, c, nulls, info = K.dense_count_many([job])[0]
        got = pd.Series(c.cpu().numpy(), index=k.cpu().numpy()).sort_index()
        assert got.index.is_unique
        np.testing.assert_array_equal(got.index.to_numpy(), e.index.to_numpy())
        np.testing.assert_array_equal(got.to_numpy(), e.to_numpy())
        assert nulls == (int(mask.sum()) if v is not None else 0)
        assert info["max_count"] >= int(e.max()) and info["rows"] == n
