"""Round-2 parity holes: LambdaOp on the device and on the host fallback
(reference calling conventions tests/unit/ops/test_lambda.py:48-50,118-120), NormalizeMinMax
against the oracle incl. the max == min branch (normalize.py:150-161), the literal-reference
tie order (categorify.py:1300,1316), to_parquet(dtypes=...) on several partitions, hashing of
null rows."""
import json
import os
import sys

import numpy as np
import pandas as pd
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _frame(n=50_000, seed=3):
    rng = np.random.default_rng(seed)
    df = pd.DataFrame({
        "a": rng.integers(0, 1000, n).astype("int32"),
        "b": rng.integers(1, 50, n).astype("int64"),
        "x": rng.normal(size=n).astype("float32"),
        "y": pd.array(rng.integers(0, 9, n), dtype="Int32"),
        "s": rng.choice(["alpha", "beta", "gamma", "delta"], n),
    })
    df.loc[rng.random(n) < 0.1, "y"] = pd.NA
    df.loc[rng.random(n) < 0.05, "x"] = np.nan
    return df


def _host_view(df):
    out = df.copy()
    for c in out.columns:
        if isinstance(out[c].dtype, pd.api.extensions.ExtensionDtype):
            out[c] = out[c].astype("float64")
    return out


@pytest.mark.parametrize("udf, cols, path, dep", [
    (lambda col: col + 100, ["a", "b"], "device", None),                # test_lambda.py:118-120
    (lambda col: col.astype(float), ["a"], "device", None),             # test_lambda.py:98-104
    (lambda col: np.log(col.fillna(0).clip(0, 5) + 1), ["x", "y"], "device", None),
    (lambda col, df: col * df["b"] - df["a"] / 4, ["a", "x"], "device", ["b"]),  # f(col, df)
    (lambda col: (col > 3) & (col < 700), ["a"], "device", None),
    (lambda col: col.str.slice(1, 3), ["s"], "host", None),              # test_lambda.py:48-50
])
def test_lambdaop_device_and_host_paths_match_pandas(udf, cols, path, dep):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops
    from nvtabular_amd.device import DeviceFrame

    df = _frame()
    op = ops.LambdaOp(udf, dependency=dep)
    wf = nvt.Workflow(cols >> op)
    frame = DeviceFrame.from_pandas(df)
    wf.fit(nvt.Dataset(frame))
    out = wf.transform(frame)
    assert op.last_path == path
    host = _host_view(df)
    for c in cols:
        from inspect import signature

        exp = udf(host[c], host) if len(signature(udf).parameters) == 2 else udf(host[c])
        got = out[c].to_pandas(c)
        if exp.dtype == object:
            assert got.tolist() == exp.tolist()
        elif exp.dtype == bool:
            np.testing.assert_array_equal(got.to_numpy().astype(bool), exp.to_numpy())
        else:
            np.testing.assert_allclose(got.to_numpy().astype("float64"), exp.to_numpy().astype("float64"),
                                       rtol=1e-6, atol=0, equal_nan=True)


def test_lambdaop_autowrap_after_categorify_stays_on_device(tmp_path):
    """test_lambda.py:118-126: Categorify >> (lambda col: col + 100)."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops
    from nvtabular_amd.workflow import iter_nodes

    df = _frame()
    node = ["a", "b"] >> ops.Categorify(out_path=str(tmp_path)) >> (lambda col: col + 100)
    wf = nvt.Workflow(node)
    out = wf.fit_transform(nvt.Dataset(df)).to_ddf().compute()
    lam = [n.op for n in iter_nodes(wf.output_node) if isinstance(n.op, ops.LambdaOp)][0]
    assert lam.last_path == "device"
    paths = O.categorify_fit([df], ["a", "b"], str(tmp_path / "cpu"), tie_break="stable")
    exp = O.categorify_transform(df, ["a", "b"], paths)
    for c in ("a", "b"):
        assert pd.api.types.is_integer_dtype(out[c].dtype)
        np.testing.assert_array_equal(out[c].to_numpy(), exp[c].to_numpy() + 100)


def test_normalize_minmax_vs_oracle_incl_constant_column():
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(5)
    n = 40_000
    df = pd.DataFrame({
        "u": rng.normal(3, 7, n),
        "v": rng.integers(-50, 50, n).astype("int32"),
        "const": np.full(n, 7.0),          # max == min -> x / (2x) = 0.5 (normalize.py:155-160)
        "zero": np.zeros(n),               # max == min == 0 -> 0 / 0 = NaN
        "w": rng.normal(0, 3, n).astype("float32"),   # float32 stays float32 in pandas
    })
    df.loc[rng.random(n) < 0.1, "u"] = np.nan
    cols = ["u", "v", "const", "zero", "w"]
    parts = [df.iloc[: n // 2].reset_index(drop=True), df.iloc[n // 2:].reset_index(drop=True)]
    op = ops.NormalizeMinMax()
    wf = nvt.Workflow(cols >> op)
    wf.fit(nvt.Dataset(df, npartitions=2))
    out = wf.transform(nvt.Dataset(df)).to_ddf().compute()
    mins, maxs = O.minmax_fit(parts, cols)
    for c in cols:
        assert op.mins[c] == float(mins[c]) and op.maxs[c] == float(maxs[c]), c
    exp = O.minmax_transform(df, cols, mins, maxs)
    for c in cols:
        np.testing.assert_allclose(out[c].to_numpy(), exp[c].to_numpy(), rtol=1e-12, atol=0, equal_nan=True)
    # float32 input: (x - min) / dif is two float32 operations in pandas and in the kernel
    np.testing.assert_array_equal(out["w"].to_numpy(), exp["w"].to_numpy())


def test_tie_break_reference_equals_literal_pandas_order(tmp_path):
    """Many equal-count categories: the literal two sort_values calls of the reference
    (oracle tie_break="pandas") and Categorify(tie_break="reference") agree label for label."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(11)
    n = 60_000
    df = pd.DataFrame({
        "c": rng.integers(0, 20_000, n).astype("int32"),       # counts 1..8: huge tie blocks
        "d": pd.array(rng.zipf(1.3, n) % 3000, dtype="Int32"),
        "s": rng.choice([f"k{i}" for i in range(500)], n),
    })
    df.loc[rng.random(n) < 0.05, "d"] = pd.NA
    cols = ["c", "d", "s"]
    wf = nvt.Workflow(cols >> ops.Categorify(out_path=str(tmp_path / "gpu"), tie_break="reference"))
    out = wf.fit_transform(nvt.Dataset(df)).to_ddf().compute()
    host = _host_view(df)
    paths = O.categorify_fit([host], cols, str(tmp_path / "cpu"), tie_break="pandas")
    exp = O.categorify_transform(host, cols, paths)
    for c in cols:
        np.testing.assert_array_equal(out[c].to_numpy(), exp[c].to_numpy(), err_msg=c)
        a = pd.read_parquet(tmp_path / "gpu" / "categories" / f"unique.{c}.parquet")
        b = pd.read_parquet(paths[c])
        assert a[c].astype(object).where(a[c].notna(), None).tolist() == \
            b[c].astype(object).where(b[c].notna(), None).tolist(), c


def test_to_parquet_dtypes_keeps_one_file_per_partition(tmp_path):
    """the dtypes cast loop clobbered the partition index."""
    import pyarrow.parquet as pq

    import nvtabular_amd as nvt

    df = _frame(40_000)[["a", "b", "x"]]
    ds = nvt.Dataset(df, npartitions=4)
    ds.to_parquet(str(tmp_path), dtypes={"a": "int64", "x": "float64"})
    files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".parquet"))
    assert len(files) == 4, files
    rows = [pq.ParquetFile(tmp_path / f).metadata.num_rows for f in files]
    assert sum(rows) == len(df) and min(rows) > 0
    back = pd.concat([pd.read_parquet(tmp_path / f) for f in files], ignore_index=True)
    assert back["a"].dtype == "int64" and back["x"].dtype == "float64"
    np.testing.assert_array_equal(back["a"].to_numpy(), df["a"].to_numpy())


def test_hash_bucket_null_rows_hash_as_key_zero_from_arrow(tmp_path):
    """Arrow leaves arbitrary bytes under a null; HashBucket / HashedCross must
    not depend on them (pandas input zero-fills, parquet input does not)."""
    import pyarrow as pa
    import pyarrow.parquet as pq

    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(2)
    n = 30_000
    vals = rng.integers(1, 10**6, n).astype("int64")
    mask = rng.random(n) < 0.2
    # values buffer with garbage under the nulls
    arr = pa.Array.from_buffers(pa.int64(), n, [pa.py_buffer(np.packbits(~mask, bitorder="little")),
                                                  pa.py_buffer(vals)], null_count=int(mask.sum()))
    other = pa.array(rng.integers(0, 100, n).astype("int32"))
    path = str(tmp_path / "in.parquet")
    pq.write_table(pa.table({"k": arr, "o": other}), path)
    wf = nvt.Workflow((["k"] >> ops.HashBucket(1000)) + ([["k", "o"]] >> ops.HashedCross(777)))
    out = wf.fit_transform(nvt.Dataset(path, engine="parquet")).to_ddf().compute()
    host = pd.DataFrame({"k": np.where(mask, np.nan, vals.astype("float64")), "o": np.asarray(other)})
    exp_b = O.hash_bucket_op(host[["k"]].copy(), 1000, ["k"])["k"]
    exp_x = O.hashed_cross(host, ["k", "o"], 777)["k_X_o"]
    np.testing.assert_array_equal(out["k"].to_numpy(), exp_b.to_numpy())
    np.testing.assert_array_equal(out["k_X_o"].to_numpy(), exp_x.to_numpy())


def test_fully_staged_encode_with_unseen_keys_and_nulls(tmp_path):
    """int32 vocabularies <= NVT_ENCODE_RESIDENT_I32 are encoded from the LDS table alone (no
    table in HBM): labels incl. out-of-vocabulary rows and nulls must equal the oracle's."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(5)
    n = 400_000
    df = pd.DataFrame({
        "a": (rng.zipf(1.2, n) % 8000).astype("int32") * 7919 - 12345,   # ~6-8 k distinct
        "b": pd.array(rng.integers(0, 3000, n), dtype="Int32"),
    })
    df.loc[rng.random(n) < 0.03, "b"] = pd.NA
    cols = ["a", "b"]
    wf = nvt.Workflow(cols >> ops.Categorify(out_path=str(tmp_path / "gpu")))
    wf.fit(nvt.Dataset(df))
    df2 = df.copy()
    df2.loc[::97, "a"] = 2_000_000_001   # never seen by fit
    out = wf.transform(nvt.Dataset(df2)).to_ddf().compute()
    host = _host_view(df)
    paths = O.categorify_fit([host], cols, str(tmp_path / "cpu"), tie_break="stable")
    exp = O.categorify_transform(_host_view(df2), cols, paths)
    for c in cols:
        np.testing.assert_array_equal(out[c].to_numpy(), exp[c].to_numpy(), err_msg=c)


@pytest.mark.parametrize("vocab", [1, 3, 1000, 1024, 1025, 2048, 2049, 4096, 4097, 8192, 8193])
@pytest.mark.parametrize("num_buckets", [None, 7])
def test_small_vocabulary_encode_at_the_table_boundaries(tmp_path, vocab, num_buckets):
    """Vocabularies <= 1024 / <= 2048 keys take encode_small_kernel (2048- / 4096-slot LDS tables,
    256-thread workgroups, two keys per lane and run of 128), up to 8192 keys the fully staged kernel,
    8193 the cache mode over a table in HBM: the labels on both sides of each boundary -- rows not a
    multiple of 128, nulls, keys the fit never saw (hashed into buckets or not), INT32_MIN as an
    ordinary key -- must equal the oracle's."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    rng = np.random.default_rng(vocab)
    n = 100_000 + 77
    ids = np.arange(vocab, dtype=np.int64) * 104729 - 5_000_000
    ids[0] = np.iinfo(np.int32).min      # the tables' empty marker is a key like any other
    x = ids[np.minimum(rng.zipf(1.3, n) - 1, vocab - 1)] if vocab > 1 else np.full(n, ids[0])
    x[:vocab] = ids                       # every key occurs
    df = pd.DataFrame({"a": pd.array(x.astype("int32"), dtype="Int32")})
    df.loc[rng.random(n) < 0.05, "a"] = pd.NA
    kw = {} if num_buckets is None else {"num_buckets": num_buckets}
    wf = nvt.Workflow(["a"] >> ops.Categorify(out_path=str(tmp_path / "gpu"), **kw))
    wf.fit(nvt.Dataset(df))
    df2 = df.copy()
    df2.loc[::53, "a"] = 1_999_999_999    # never seen by fit
    df2.loc[5::997, "a"] = -1_999_999_999
    out = wf.transform(nvt.Dataset(df2)).to_ddf().compute()
    paths = O.categorify_fit([_host_view(df)], ["a"], str(tmp_path / "cpu"), tie_break="stable", **kw)
    exp = O.categorify_transform(_host_view(df2), ["a"], paths, **kw)
    np.testing.assert_array_equal(out["a"].to_numpy(), exp["a"].to_numpy())


def test_multi_partition_fit_on_the_filtered_partitioned_path(tmp_path):
    """Three partitions of a 150 k-key power-law column: every partition is counted on a
    partitioned path behind its own sampled hot set (the range path, or path 1 | NVT_PATH_HOT
    with NVT_RANGE=0), the per-partition lists are merged by the
    weighted count -- vocabulary and labels equal the oracle's on the whole frame."""
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K, ops

    rng = np.random.default_rng(21)
    n = 1_200_000
    df = pd.DataFrame({
        "c": pd.array((rng.zipf(1.08, n) % 150_000).astype("int32") * 13 - 99, dtype="Int32"),
    })
    df.loc[rng.random(n) < 0.02, "c"] = pd.NA
    parts = [df.iloc[i * 400_000:(i + 1) * 400_000].reset_index(drop=True) for i in range(3)]
    host = _host_view(df)
    # (a partitioned path: the range path by default, the filtered hash path 1 with NVT_RANGE=0)
    assert K._path_for(host["c"].nunique()) in (1, K.PATH_RANGE) and K.HOT_FILTER
    wf = nvt.Workflow(["c"] >> ops.Categorify(out_path=str(tmp_path / "gpu")))
    wf.fit(nvt.Dataset(parts))
    out = wf.transform(nvt.Dataset(df)).to_ddf().compute()
    paths = O.categorify_fit([_host_view(p) for p in parts], ["c"], str(tmp_path / "cpu"),
                             tie_break="stable")
    exp = O.categorify_transform(host, ["c"], paths)
    np.testing.assert_array_equal(out["c"].to_numpy(), exp["c"].to_numpy())
    a = pd.read_parquet(tmp_path / "gpu" / "categories" / "unique.c.parquet")
    b = pd.read_parquet(paths["c"])
    np.testing.assert_array_equal(a["c"].to_numpy(), b["c"].to_numpy())
    np.testing.assert_array_equal(a["c_size"].to_numpy(), b["c_size"].to_numpy())
