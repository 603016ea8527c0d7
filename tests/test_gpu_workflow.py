"""Workflow-level parity on the GPU: parquet datasets, list columns, joint groups of mixed
dtypes, user vocabularies, single_table, save / load -- against the oracle or the
reference's own expectations (file:line cited)."""
import os

import numpy as np
import pandas as pd
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _criteo_like(n, seed=0):
    rng = np.random.default_rng(seed)
    df = pd.DataFrame({
        "C1": pd.array(rng.zipf(1.2, n) % 5000, dtype="Int32"),
        "C2": pd.array(rng.integers(0, 40, n), dtype="Int32"),
        "C3": rng.integers(-2**40, 2**40, n) // 2**28,  # int64, ~8k distinct, negative too
        "I1": pd.array(np.floor(rng.lognormal(2, 2, n)).astype("int64"), dtype="Int32"),
        "I2": rng.normal(size=n).astype("float32"),
        "label": rng.integers(0, 2, n).astype("int32"),
    })
    for c, f in (("C1", 0.1), ("C2", 0.02), ("I1", 0.3)):
        df.loc[rng.random(n) < f, c] = pd.NA
    df.loc[rng.random(n) < 0.05, "I2"] = np.nan
    return df


def _oracle_view(df):
    """What pandas' default parquet reader hands the reference: nullable ints -> float64."""
    out = df.copy()
    for c in out.columns:
        if isinstance(out[c].dtype, pd.api.extensions.ExtensionDtype):
            out[c] = out[c].astype("float64")
    return out


def test_parquet_dataset_fit_transform_vs_oracle(tmp_path):
    """cfg1/cfg2 shape end to end: parquet row groups -> Arrow buffers -> HBM -> ops -> pandas;
    3 files x 2 row groups = 6 partitions (tree merge + per-partition transform)."""
    import pyarrow as pa
    import pyarrow.parquet as pq

    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    df = _criteo_like(60_000, seed=1)
    paths = []
    for i, part in enumerate(np.array_split(np.arange(len(df)), 3)):
        p = str(tmp_path / f"day_{i}.parquet")
        pq.write_table(pa.Table.from_pandas(df.iloc[part], preserve_index=False), p,
                       row_group_size=10_000)
        paths.append(p)
    cats = ["C1", "C2", "C3"] >> ops.Categorify(out_path=str(tmp_path / "gpu"))
    conts = ["I1", "I2"] >> ops.FillMissing() >> ops.Normalize()
    wf = nvt.Workflow(cats + conts + ["label"])
    ds = nvt.Dataset(paths, engine="parquet")
    assert ds.npartitions == 6
    wf.fit(ds)
    got = wf.transform(ds).to_ddf().compute()

    odf = _oracle_view(df)
    parts = [odf.iloc[i : i + 10_000].reset_index(drop=True) for i in range(0, len(odf), 10_000)]
    cpaths = O.categorify_fit(parts, ["C1", "C2", "C3"], str(tmp_path / "cpu"), tie_break="stable")
    exp = O.categorify_transform(odf, ["C1", "C2", "C3"], cpaths)
    for c in ("C1", "C2", "C3"):
        assert got[c].dtype == np.int64
        np.testing.assert_array_equal(got[c].to_numpy(), exp[c].to_numpy())
    filled = [O.fill_missing(p[["I1", "I2"]].copy(), ["I1", "I2"], 0) for p in parts]
    mom = O.custom_moments(filled, ["I1", "I2"])
    op = [n for n in [wf.output_node] for _ in [0]][0]
    ofull = O.fill_missing(odf[["I1", "I2"]].copy(), ["I1", "I2"], 0)
    ref = O.normalize_transform(ofull, ["I1", "I2"], mom["mean"].to_dict(), mom["std"].to_dict())
    np.testing.assert_allclose(got["I1"].to_numpy(), ref["I1"].to_numpy(), rtol=1e-6, atol=1e-9)
    # float32 input: pandas subtracts and divides in float32 (normalize.py:79-84), and so does
    # the kernel -- the same two correctly rounded operations, bit for bit GIVEN the same
    # mean / std.  (The statistics themselves agree to 1e-6, not to the last bit: pandas sums a
    # float32 column differently; one ulp in float32(std) moves ~7 % of the quotients by an ulp.)
    from nvtabular_amd.node import iter_nodes

    norm = [n.op for n in iter_nodes(wf.output_node) if type(n.op).__name__ == "Normalize"][0]
    for c in ("I1", "I2"):
        assert abs(norm.means[c] - mom["mean"][c]) <= 1e-6 * max(1.0, abs(mom["mean"][c]))
        assert abs(norm.stds[c] - mom["std"][c]) <= 1e-6 * mom["std"][c]
    np.testing.assert_allclose(got["I2"].to_numpy(), ref["I2"].to_numpy(), rtol=1e-5, atol=1e-7)
    ref_same = O.normalize_transform(ofull, ["I2"], norm.means, norm.stds)
    np.testing.assert_array_equal(got["I2"].to_numpy(), ref_same["I2"].to_numpy())
    np.testing.assert_array_equal(got["label"].to_numpy(), df["label"].to_numpy())
    # schema: test_categorify.py:532-540 / categorify.py:564-577
    props = wf.output_schema["C2"].properties
    card = len(pd.read_parquet(cpaths["C2"])) + 3
    assert props["embedding_sizes"]["cardinality"] == card and props["domain"]["max"] == card - 1


def test_list_columns_int_categorify_and_hashbucket(tmp_path):
    """cfg5 shape: multi-hot list<int> column; Categorify and HashBucket act on the leaves and
    keep the offsets (categorify.py:1696,1803; hash_bucket.py:93-96)."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(5)
    n = 5000
    lens = rng.integers(0, 6, n)
    rows = [rng.zipf(1.3, k).astype("int64") % 300 for k in lens]
    df = pd.DataFrame({"tags": rows, "item": rng.integers(0, 50, n)})
    wf = nvt.Workflow(["tags", "item"] >> ops.Categorify(out_path=str(tmp_path / "g"), freq_threshold=3))
    got = wf.fit_transform(nvt.Dataset(df)).to_ddf().compute()
    paths = O.categorify_fit([df], ["tags", "item"], str(tmp_path / "c"), freq_threshold=3,
                             tie_break="stable")
    exp = O.categorify_transform(df, ["tags", "item"], paths)
    assert [len(r) for r in got["tags"]] == lens.tolist()
    np.testing.assert_array_equal(np.concatenate(got["tags"].to_list()),
                                  np.concatenate(exp["tags"].to_list()))
    np.testing.assert_array_equal(got["item"].to_numpy(), exp["item"].to_numpy())
    hb = nvt.Workflow(["tags"] >> ops.HashBucket(64)).transform(df)
    ehb = O.hash_bucket_op(df.copy(), 64, cols=["tags"])
    np.testing.assert_array_equal(np.concatenate(hb["tags"].to_list()),
                                  np.concatenate(ehb["tags"].to_list()))
    assert hb["tags"].iloc[int(np.argmax(lens))].dtype == np.int32


def test_joint_group_mixed_dtypes_and_lists(tmp_path):
    """test_categorify.py:636-665 with integers: a scalar int32 column and a list<int64>
    column sharing one vocabulary."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    df = pd.DataFrame({
        "Author": np.array([10, 50, 20, 30], dtype="int32"),
        "Engaging": [np.array([20, 30], dtype="int64"), np.array([], dtype="int64"),
                     np.array([10, 40], dtype="int64"), np.array([10], dtype="int64")],
    })
    cats = [["Author", "Engaging"]] >> ops.Categorify(out_path=str(tmp_path))
    out = nvt.Workflow(cats).fit_transform(nvt.Dataset(df)).to_ddf().compute()
    assert out["Author"].tolist() == [3, 7, 4, 5]
    assert np.concatenate(out["Engaging"].to_list()).tolist() == [4, 5, 3, 6, 3]


def test_user_vocabs_and_single_table(tmp_path):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    # test_categorify.py:123-157 (vocabs branch): given order is kept, labels start at 3
    df = pd.DataFrame({"Authors": [["User_A"], ["User_A", "User_E"], ["User_B", "User_C"], ["User_C"]]})
    vocabs = {"Authors": pd.Series([f"User_{x}" for x in "ACBE"])}
    out = nvt.Workflow(["Authors"] >> ops.Categorify(out_path=str(tmp_path / "v"), vocabs=vocabs)) \
        .fit_transform(nvt.Dataset(df)).to_ddf().compute()
    assert [list(r) for r in out["Authors"]] == [[3], [3, 6], [5, 4], [4]]
    # test_categorify.py:509-529: single_table gives monotone, non-overlapping ranges
    df2 = pd.DataFrame({
        "Authors": [None, "User_A", "User_A", "User_E", "User_B", "User_C"],
        "Engaging_User": [None, "User_B", "User_B", "User_A", "User_D", "User_D"],
    })
    wf = nvt.Workflow(["Authors", "Engaging_User"] >> ops.Categorify(out_path=str(tmp_path / "s"),
                                                                     single_table=True))
    new = wf.fit_transform(nvt.Dataset(df2)).to_ddf().compute()
    old_max = 1
    for name in ["Authors", "Engaging_User"]:
        assert old_max <= new[name].min()
        old_max += new[name].max()


def test_vocabs_dict_of_paths_reuses_fitted_files(tmp_path):
    """categorify.py:441-446: `vocabs={col: "path/to/unique.col.parquet"}` -- the files of an
    earlier fit are taken as they are (no fit for those columns), labels equal the fit's."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    df = _criteo_like(20_000, seed=7)
    wf1 = nvt.Workflow(["C1", "C2"] >> ops.Categorify(out_path=str(tmp_path / "first")))
    exp = wf1.fit_transform(nvt.Dataset(df)).to_ddf().compute()
    paths = {c: str(tmp_path / "first" / "categories" / f"unique.{c}.parquet") for c in ("C1", "C2")}
    assert all(os.path.exists(p) for p in paths.values())
    op = ops.Categorify(out_path=str(tmp_path / "second"), vocabs=paths)
    assert op.categories == paths
    wf2 = nvt.Workflow(["C1", "C2"] >> op)
    got = wf2.fit_transform(nvt.Dataset(df)).to_ddf().compute()
    for c in ("C1", "C2"):
        np.testing.assert_array_equal(got[c].to_numpy(), exp[c].to_numpy())
    # nothing was fitted or rewritten for the supplied columns
    assert not os.path.exists(tmp_path / "second" / "categories" / "unique.C1.parquet")
    # a vocabulary for one column only: the other one is fitted as usual
    op3 = ops.Categorify(out_path=str(tmp_path / "third"), vocabs={"C1": paths["C1"]})
    got3 = nvt.Workflow(["C1", "C2"] >> op3).fit_transform(nvt.Dataset(df)).to_ddf().compute()
    for c in ("C1", "C2"):
        np.testing.assert_array_equal(got3[c].to_numpy(), exp[c].to_numpy())
    with pytest.raises(ValueError, match="Unrecognized vocab type"):
        ops.Categorify(vocabs={"C1": 3})


@pytest.mark.parametrize("max_size", [0, 6])
@pytest.mark.parametrize("buckets", [None, 3])
def test_placement_knobs_do_not_change_the_result(tmp_path, max_size, buckets):
    """test_categorify.py:668-704 (split_out) and categorify.py:152-173: split_out / on_host /
    cat_cache / search_sorted steer where the reference keeps and how it searches the
    categories; labels and the unique.*.parquet contents must not depend on them."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    df = pd.DataFrame({"user_id": np.array([1, 2, 3, 4, 6, 8, 5, 3] * 10, dtype="int64")})
    kw = dict(max_size=max_size, num_buckets=buckets)
    wf1 = nvt.Workflow(["user_id"] >> ops.Categorify(out_path=str(tmp_path / "a"), split_out=1, **kw))
    r1 = wf1.fit_transform(nvt.Dataset(df)).to_ddf().compute()
    wfn = nvt.Workflow(["user_id"] >> ops.Categorify(
        out_path=str(tmp_path / "b"), split_out=2, split_every=4, on_host=False,
        cat_cache="device", search_sorted=True, **kw))
    rn = wfn.fit_transform(nvt.Dataset(df, npartitions=3)).to_ddf().compute()
    np.testing.assert_array_equal(rn["user_id"].to_numpy(), r1["user_id"].to_numpy())
    c1 = pd.read_parquet(tmp_path / "a" / "categories" / "unique.user_id.parquet")
    cn = pd.read_parquet(tmp_path / "b" / "categories" / "unique.user_id.parquet")
    pd.testing.assert_frame_equal(c1, cn)
    with pytest.warns(FutureWarning):
        ops.Categorify(tree_width=8)
    with pytest.raises(ValueError):
        ops.Categorify(search_sorted=True, freq_threshold=2)


def test_save_load_roundtrip_and_eager_artifacts(tmp_path):
    """tests/unit/workflow/test_workflow.py:691-759: a saved workflow transforms identically
    after load (encoders rebuilt from unique.*.parquet), stats files live under artifacts/."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    df = _criteo_like(20_000, seed=3)
    cats = ["C1", "C3"] >> ops.Categorify(out_path=str(tmp_path / "stats"))
    conts = ["I1"] >> ops.FillMissing(fill_val=7) >> ops.Normalize()
    jg = ["C2"] >> ops.JoinGroupby(cont_cols=["I2"], stats=["mean", "sum"], out_path=str(tmp_path / "stats"))
    wf = nvt.Workflow(cats + conts + jg)
    a = wf.fit_transform(nvt.Dataset(df)).to_ddf().compute()
    assert os.path.exists(tmp_path / "stats" / "categories" / "unique.C1.parquet")
    wf.save(str(tmp_path / "saved"))
    assert os.path.isdir(tmp_path / "saved" / "artifacts")
    wf2 = nvt.Workflow.load(str(tmp_path / "saved"))
    b = wf2.transform(nvt.Dataset(df)).to_ddf().compute()
    pd.testing.assert_frame_equal(a, b)


def test_target_encoding_parquet_partitions_fold_alignment(tmp_path):
    """Folds are re-seeded per partition (target_encoding.py:427-439): identical partition
    boundaries in fit and transform give the oracle's values."""
    import pyarrow as pa
    import pyarrow.parquet as pq

    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(8)
    n = 30_000
    df = pd.DataFrame({"k": rng.integers(0, 200, n).astype("int32"),
                       "y": rng.random(n).astype("float32")})
    p = str(tmp_path / "d.parquet")
    pq.write_table(pa.Table.from_pandas(df, preserve_index=False), p, row_group_size=8_000)
    ds = nvt.Dataset(p)
    te = ["k"] >> ops.TargetEncoding("y", kfold=5, fold_seed=42, p_smooth=20,
                                     out_path=str(tmp_path / "g"))
    got = nvt.Workflow(te).fit_transform(ds).to_ddf().compute()
    parts = [df.iloc[i : i + 8_000].reset_index(drop=True) for i in range(0, n, 8_000)]
    stats, means = O.target_encoding_fit([q.copy() for q in parts], ["k"], ["y"], str(tmp_path / "c"),
                                         kfold=5, fold_seed=42)
    exp = pd.concat([O.target_encoding_transform(q.copy(), ["k"], ["y"], stats, means, kfold=5,
                                                 fold_seed=42, p_smooth=20) for q in parts],
                    ignore_index=True)
    np.testing.assert_allclose(got["TE_k_y"].to_numpy(), exp["TE_k_y"].to_numpy(), rtol=1e-5, atol=1e-6)


def test_parquet_prefetch_side_stream_equals_synchronous(tmp_path):
    """pinned staging + async copies on a side stream must not change
    results."""
    import pyarrow as pa
    import pyarrow.parquet as pq

    import nvtabular_amd as nvt

    df = _criteo_like(40_000, seed=9)
    p = str(tmp_path / "d.parquet")
    pq.write_table(pa.Table.from_pandas(df, preserve_index=False), p, row_group_size=5_000)
    ds = nvt.Dataset(p)
    a = pd.concat([f.to_pandas() for f in ds.to_iter(prefetch=True)], ignore_index=True)
    b = pd.concat([f.to_pandas() for f in ds.to_iter(prefetch=False)], ignore_index=True)
    pd.testing.assert_frame_equal(a, b)
    pd.testing.assert_frame_equal(a[["C3", "label"]], df[["C3", "label"]].reset_index(drop=True))
    np.testing.assert_array_equal(a["C1"].isna().to_numpy(), df["C1"].isna().to_numpy())


@pytest.mark.parametrize("shuffle", ["per-worker", "per-partition", None])
def test_parquet_output_contract(tmp_path, shuffle):
    """tests/unit/workflow/test_workflow.py:363-396 (test_parquet_output) and :444-500
    (test_workflow_apply) of the reference: file count, _metadata, dtypes."""
    import glob
    import os

    import pyarrow.parquet as pq

    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    assert nvt.io.Shuffle.PER_WORKER == "per-worker"
    size, k = 25, 2
    path = str(tmp_path / "simple.parquet")
    pd.DataFrame({"a": np.arange(size)}).to_parquet(path, row_group_size=5, engine="pyarrow")
    ds = nvt.Dataset(path, engine="parquet", row_groups_per_part=1)
    out = str(tmp_path / "processed")
    wf = nvt.Workflow(["a"] >> ops.Normalize())
    wf.fit_transform(ds).to_parquet(output_path=out, shuffle=shuffle, out_files_per_proc=k)
    files = glob.glob(os.path.join(out, "*.parquet"))
    assert len(files) == k
    md = pq.read_metadata(os.path.join(out, "_metadata"))
    assert md.num_rows == size and md.schema.names == ["a"]
    got = pd.concat([pd.read_parquet(f) for f in sorted(files)])["a"].to_numpy()
    exp = (np.arange(size) - np.arange(size).mean()) / np.arange(size).std(ddof=1)
    np.testing.assert_allclose(np.sort(got), np.sort(exp), rtol=1e-6, atol=1e-9)
    if shuffle is None:
        np.testing.assert_allclose(pd.read_parquet(sorted(files)[0])["a"].to_numpy()[:2], exp[:2],
                                   rtol=1e-6, atol=1e-9)
    assert open(os.path.join(out, "_file_list.txt")).read().split()[0] == str(k)
    # round trip through Dataset
    back = nvt.Dataset(files).to_ddf().compute()
    assert len(back) == size

    # forced dtypes (test_workflow_apply)
    df = pd.DataFrame({
        "cont1": np.arange(size, dtype=np.float64), "cont2": np.arange(size, dtype=np.float64),
        "cat1": np.arange(size, dtype=np.int32), "cat2": np.arange(size, dtype=np.int32),
        "label": np.arange(size, dtype=np.float64)})
    p2 = str(tmp_path / "five.parquet")
    df.to_parquet(p2, row_group_size=5, engine="pyarrow")
    ds2 = nvt.Dataset(p2, engine="parquet", row_groups_per_part=1)
    cats = ["cat1", "cat2"] >> ops.Categorify(out_path=str(tmp_path / "cats"))
    conts = ["cont1", "cont2"] >> ops.FillMissing() >> ops.Clip(min_value=0) >> ops.LogOp
    wf2 = nvt.Workflow(cats + conts + ["label"]).fit(ds2)
    want = {"cont1": np.float32, "cont2": np.float32, "cat1": np.float32, "cat2": np.float32,
            "label": np.int64}
    out2 = str(tmp_path / "processed2")
    wf2.transform(ds2).to_parquet(output_path=out2, shuffle=shuffle, out_files_per_proc=k,
                                  dtypes=want, cats=["cat1", "cat2"], conts=["cont1", "cont2"],
                                  labels=["label"])
    for f in glob.glob(os.path.join(out2, "*.parquet")):
        g = pd.read_parquet(f)
        assert {c: g[c].dtype for c in g.columns} == {c: np.dtype(t) for c, t in want.items()}
    import json

    meta = json.load(open(os.path.join(out2, "_metadata.json")))
    assert sum(x["num_rows"] for x in meta["file_stats"]) == size
    assert [c["col_name"] for c in meta["cats"]] == ["cat1", "cat2"]


@pytest.mark.parametrize("k", [None, 3])
def test_plain_parquet_writer_equals_pyarrow_path(tmp_path, k, monkeypatch):
    """Dataset.to_parquet through the hand-written PLAIN writer (nullable int32 / float columns,
    row groups smaller than a partition, file splits that do not fall on byte boundaries of the
    validity bitmaps) reads back equal to the same dataset written by pyarrow."""
    import glob
    import os

    import nvtabular_amd as nvt
    from nvtabular_amd import io as nio

    df = _criteo_like(30_011, seed=4)
    num = [c for c in df.columns if df[c].dtype.kind in "if"]
    parts = [df[num].iloc[:17_003].reset_index(drop=True), df[num].iloc[17_003:].reset_index(drop=True)]
    monkeypatch.setattr(nio, "PLAIN_ROW_GROUP", 4096)
    outs = {}
    for mode in ("plain", "arrow"):
        monkeypatch.setattr(nio, "PLAIN_PARQUET", mode == "plain")
        out = str(tmp_path / mode)
        nio.LAST_TIMING.clear()
        nvt.Dataset([p.copy() for p in parts]).to_parquet(out, out_files_per_proc=k)
        assert bool(nio.LAST_TIMING) == (mode == "plain")          # the plain path really ran
        files = sorted(glob.glob(os.path.join(out, "*.parquet")))
        assert len(files) == (k or len(parts))
        outs[mode] = pd.concat([pd.read_parquet(f) for f in files], ignore_index=True)
        assert open(os.path.join(out, "_file_list.txt")).read().split()[0] == str(len(files))
    pd.testing.assert_frame_equal(outs["plain"], outs["arrow"])
    assert outs["plain"][num[0]].isna().sum() == df[num[0]].isna().sum()


def test_to_parquet_statistics_of_the_plain_writer(tmp_path, monkeypatch):
    """to_parquet(statistics=True): min / max of every chunk computed on the device next to the
    copy out, null counts always -- equal to what pyarrow computes for the same data."""
    import glob
    import os

    import pyarrow.parquet as pq

    import nvtabular_amd as nvt
    from nvtabular_amd import io as nio

    df = _criteo_like(20_011, seed=6)
    num = [c for c in df.columns if df[c].dtype.kind in "if"]
    monkeypatch.setattr(nio, "PLAIN_ROW_GROUP", 8192)
    out = str(tmp_path / "s")
    nio.LAST_TIMING.clear()
    nvt.Dataset(df[num].copy()).to_parquet(out, statistics=True)
    assert nio.LAST_TIMING   # the plain path ran
    f = sorted(glob.glob(os.path.join(out, "*.parquet")))[0]
    md = pq.read_metadata(f)
    back = pd.read_parquet(f)
    table = pq.read_table(f)
    at = 0
    for g in range(md.num_row_groups):
        rows = md.row_group(g).num_rows
        part = back.iloc[at:at + rows]
        for j, c in enumerate(md.schema.names):
            st = md.row_group(g).column(j).statistics
            col = part[c]
            # (parquet nulls: a NaN of a float column is a value, not a null)
            assert st.null_count == table.column(c).slice(at, rows).null_count, c
            vals = col.dropna().to_numpy()
            vals = vals[~np.isnan(vals.astype("float64"))]
            if vals.size:
                assert st.has_min_max and st.min == vals.min() and st.max == vals.max(), c
        at += rows
    # without the switch: null counts only
    out2 = str(tmp_path / "n")
    nvt.Dataset(df[num].copy()).to_parquet(out2)
    st = pq.read_metadata(sorted(glob.glob(os.path.join(out2, "*.parquet")))[0]).row_group(0).column(0).statistics
    assert st is not None and not st.has_min_max
    assert st.null_count == table.column(md.schema.names[0]).slice(0, md.row_group(0).num_rows).null_count


def test_save_load_graph_json_all_ops(tmp_path):
    """graph.json round trip through every serialisable operator of the path: the loaded
    workflow (fresh operator objects, state from JSON + artifacts/) transforms identically."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(5)
    n = 5000
    df = pd.DataFrame({
        "a": rng.integers(0, 50, n).astype("int32"), "b": rng.integers(0, 9, n),
        "x": rng.normal(size=n), "z": rng.exponential(3.0, size=n).astype("float32"),
        "y": (rng.random(n) < 0.3).astype("float32")})
    df.loc[rng.random(n) < 0.1, "x"] = np.nan
    out = str(tmp_path / "s")
    g = (["a", "b"] >> ops.Categorify(out_path=out, freq_threshold=2)) \
        + (["x"] >> ops.FillMissing() >> ops.NormalizeMinMax() >> ops.Rename(postfix="_mm")) \
        + (["z"] >> ops.Clip(min_value=0.5, max_value=9) >> ops.LogOp() >> ops.Rename(name="lz")) \
        + (["a"] >> ops.HashBucket(13) >> ops.Rename(postfix="_h")) \
        + (["a"] >> ops.TargetEncoding("y", kfold=1, p_smooth=5, out_path=out)) \
        + (["b"] >> ops.JoinGroupby(cont_cols=["x"], stats=["count", "mean"], out_path=out)) + ["y"]
    wf = nvt.Workflow(g)
    a = wf.fit_transform(nvt.Dataset(df)).to_ddf().compute()
    wf.save(str(tmp_path / "saved"))
    assert os.path.exists(tmp_path / "saved" / "graph.json")
    assert not os.path.exists(tmp_path / "saved" / "workflow.pkl")
    wf2 = nvt.Workflow.load(str(tmp_path / "saved"))
    b = wf2.transform(nvt.Dataset(df)).to_ddf().compute()
    pd.testing.assert_frame_equal(a, b[a.columns.tolist()])
    assert wf2.output_schema["a"].properties == wf.output_schema["a"].properties


def test_cfg1_movielens_shape_vs_oracle(tmp_path):
    """BASELINE.json configs[0] (MovieLens-25M: userId / movieId int categoricals, rating
    float), synthesised at 2 M rows with the real id counts (162 541 users, 59 047 movies) so
    the pandas oracle finishes in seconds: Categorify indices bit-exact, Normalize mean / std
    within 1e-6 relative."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(25)
    n = 2_000_000
    df = pd.DataFrame({
        "userId": np.minimum(rng.zipf(1.4, n), 162_541).astype("int64"),
        "movieId": (np.minimum(rng.zipf(1.2, n), 59_047) * 7 % 209_171).astype("int64"),
        "rating": (rng.integers(1, 11, n) * 0.5).astype("float32"),
    })
    wf = nvt.Workflow((["userId", "movieId"] >> ops.Categorify(out_path=str(tmp_path / "g")))
                      + (["rating"] >> ops.Normalize()))
    got = wf.fit_transform(nvt.Dataset(df)).to_ddf().compute()
    paths = O.categorify_fit([df], ["userId", "movieId"], str(tmp_path / "c"), tie_break="stable")
    exp = O.categorify_transform(df, ["userId", "movieId"], paths)
    for c in ("userId", "movieId"):
        np.testing.assert_array_equal(got[c].to_numpy(), exp[c].to_numpy())
    mom = O.custom_moments([df[["rating"]]], ["rating"])
    norm = next(n_.op for n_ in nvt.workflow.iter_nodes(wf.output_node) if isinstance(n_.op, ops.Normalize))
    assert abs(norm.means["rating"] - mom["mean"]["rating"]) <= 1e-6 * abs(mom["mean"]["rating"])
    assert abs(norm.stds["rating"] - mom["std"]["rating"]) <= 1e-6 * abs(mom["std"]["rating"])
    ref = O.normalize_transform(df[["rating"]].copy(), ["rating"], mom["mean"].to_dict(), mom["std"].to_dict())
    np.testing.assert_allclose(got["rating"].to_numpy(), ref["rating"].to_numpy(), rtol=1e-5, atol=1e-6)


def test_frame_to_arrow_keeps_types_nulls_and_lists():
    """DeviceFrame.to_arrow (the output half of the parquet path): integer columns keep their
    type and their nulls, floats their NaNs, list columns their offsets."""
    import pyarrow as pa

    from nvtabular_amd.device import DeviceFrame

    t = pa.table({
        "i": pa.array([1, None, 3, 4, None, 6, 7, 8, 9], type=pa.int32()),
        "l": pa.array([5, 6, 7, 8, 9, 10, 11, 12, 13], type=pa.int64()),
        "f": pa.array([0.5, float("nan"), 2.5, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0], type=pa.float64()),
        "lst": pa.array([[1, 2], [], [3], [4, 5, 6], [], [7], [8], [9], [10]], type=pa.list_(pa.int64())),
    })
    back = DeviceFrame.from_arrow(t).to_arrow()
    assert back.column("i").type == pa.int32() and back.column("i").to_pylist() == t.column("i").to_pylist()
    assert back.column("l").to_pylist() == t.column("l").to_pylist()
    f = back.column("f").to_pylist()
    assert f[0] == 0.5 and f[1] != f[1] and f[2:] == t.column("f").to_pylist()[2:]
    assert back.column("lst").to_pylist() == t.column("lst").to_pylist()


@pytest.mark.parametrize("ascending", [True, False])
def test_groupby_vs_oracle(ascending):
    """ops.Groupby (groupby.py:113-261; reference test tests/unit/ops/test_groupyby.py:27-110):
    list / first / last in sort order, conventional aggregations, null keys dropped."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ColumnSelector, ops

    rng = np.random.default_rng(3)
    n = 5000
    df = pd.DataFrame({
        "name": rng.choice(["Dave", "Zelda", "Ann", "Bo"], n),
        "id": rng.integers(0, 40, n).astype("int64"),
        "ts": rng.permutation(n).astype("int64"),          # unique: the sort is unambiguous
        "x": rng.integers(0, 1000, n).astype("int64"),
        "y": rng.normal(size=n),
    })
    df.loc[rng.random(n) < 0.1, "y"] = np.nan
    kdf = df.copy()
    kdf["id"] = pd.array(df["id"], dtype="Int64")
    kdf.loc[rng.random(n) < 0.05, "id"] = pd.NA            # null keys are dropped
    odf = kdf.copy()
    odf["id"] = odf["id"].astype("float64")
    aggs = {"x": ["list", "sum", "first", "last"], "y": ["first", "last", "mean", "count", "std"],
            "ts": ["min", "max"]}
    sel = ["name", "id", "ts", "x", "y"]
    feats = ColumnSelector(sel) >> ops.Groupby(groupby_cols=["name", "id"], sort_cols=["ts"],
                                               aggs=aggs, name_sep="-", ascending=ascending)
    got = nvt.Workflow(feats).fit_transform(nvt.Dataset(kdf)).to_ddf().compute()
    exp = O.groupby_op(odf, sel, ["name", "id"], ["ts"], aggs, "-", ascending)
    assert sorted(got.columns) == sorted(exp.columns) and len(got) == len(exp)
    assert got["name"].tolist() == exp["name"].tolist()
    np.testing.assert_array_equal(got["id"].to_numpy().astype("int64"), exp["id"].to_numpy().astype("int64"))
    for g, e in zip(got["x-list"], exp["x-list"]):
        assert list(g) == list(e)
    for c in ("x-first", "x-last", "ts-min", "ts-max", "y-count"):
        np.testing.assert_array_equal(got[c].to_numpy().astype("int64"), exp[c].to_numpy().astype("int64"), err_msg=c)
    assert got["y-count"].dtype == np.int32 and got["x-sum"].dtype == np.float32
    for c in ("x-sum", "y-mean", "y-std", "y-first", "y-last"):
        np.testing.assert_allclose(got[c].to_numpy().astype("float64"), exp[c].to_numpy().astype("float64"),
                                   rtol=1e-5, atol=1e-6, equal_nan=True, err_msg=c)
