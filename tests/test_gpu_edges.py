"""Edge cases of the hot path (empty / all-null / constant / extreme inputs) against the oracle."""
import numpy as np
import pandas as pd
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _cat_roundtrip(tmp_path, df, cols, **kw):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    wf = nvt.Workflow(cols >> ops.Categorify(out_path=str(tmp_path / "g"), **kw))
    got = wf.fit_transform(nvt.Dataset(df)).to_ddf().compute()
    paths = O.categorify_fit([df], cols, str(tmp_path / "c"), tie_break="stable", **kw)
    exp = O.categorify_transform(df, cols, paths)
    for c in cols:
        np.testing.assert_array_equal(got[c].to_numpy(), exp[c].to_numpy(), err_msg=c)
    return wf, got


def test_all_null_and_constant_columns(tmp_path):
    n = 1000
    df = pd.DataFrame({
        "allnull": pd.array([None] * n, dtype="Int64"),
        "const": np.full(n, 7, dtype="int64"),
        "minmax": np.where(np.arange(n) % 2 == 0, np.iinfo("int64").min, np.iinfo("int64").max),
        "i32ext": np.where(np.arange(n) % 3 == 0, np.iinfo("int32").min, np.iinfo("int32").max).astype("int32"),
    })
    odf = df.copy()
    odf["allnull"] = np.nan
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    cols = ["allnull", "const", "minmax", "i32ext"]
    wf = nvt.Workflow(cols >> ops.Categorify(out_path=str(tmp_path / "g")))
    got = wf.fit_transform(nvt.Dataset(df)).to_ddf().compute()
    # the reference writes no unique file for a column whose only group is the null group
    # (categorify.py:759-760 skips empty parts) and then cannot encode it; the engine encodes
    # every row of such a column as null (1) -- checked directly, the other columns vs the oracle
    ocols = cols[1:]
    paths = O.categorify_fit([odf], ocols, str(tmp_path / "c"), tie_break="stable")
    exp = O.categorify_transform(odf, ocols, paths)
    for c in ocols:
        np.testing.assert_array_equal(got[c].to_numpy(), exp[c].to_numpy(), err_msg=c)
    assert (got["allnull"] == 1).all() and (got["const"] == 3).all()
    meta = pd.read_parquet(tmp_path / "g" / "categories" / "meta.allnull.parquet")
    assert meta["num_observed"].tolist() == [0, n, 0, 0]


@pytest.mark.parametrize("n", [1, 2, 7, 9, 63, 65])
def test_tiny_frames_and_odd_lengths(tmp_path, n):
    rng = np.random.default_rng(n)
    df = pd.DataFrame({"a": rng.integers(0, 4, n).astype("int32"), "x": rng.normal(size=n)})
    _cat_roundtrip(tmp_path, df, ["a"])
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    if n > 1:
        wf = nvt.Workflow(["x"] >> ops.FillMissing() >> ops.Normalize())
        got = wf.fit_transform(nvt.Dataset(df)).to_ddf().compute()
        mom = O.custom_moments([df[["x"]]], ["x"])
        ref = O.normalize_transform(df[["x"]].copy(), ["x"], mom["mean"].to_dict(), mom["std"].to_dict())
        np.testing.assert_allclose(got["x"].to_numpy(), ref["x"].to_numpy(), rtol=1e-9, atol=1e-12)


def test_unseen_keys_and_nulls_at_transform_time(tmp_path):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    # (a null in the fit data too: with an all-integer vocabulary file the reference's
    # astype(vocab dtype) at categorify.py:1713 cannot take a NaN at transform time)
    fit = pd.DataFrame({"a": pd.array([5, 5, 9, 11, 11, 11, None], dtype="Int64")})
    new = pd.DataFrame({"a": pd.array([11, 1234, None, 5, -7], dtype="Int64")})
    wf = nvt.Workflow(["a"] >> ops.Categorify(out_path=str(tmp_path / "g"))).fit(nvt.Dataset(fit))
    got = wf.transform(nvt.Dataset(new)).to_ddf().compute()["a"].tolist()
    ofit = pd.DataFrame({"a": [5, 5, 9, 11, 11, 11, np.nan]})
    paths = O.categorify_fit([ofit], ["a"], str(tmp_path / "c"), tie_break="stable")
    onew = pd.DataFrame({"a": [11, 1234, np.nan, 5, -7]})
    exp = O.categorify_transform(onew, ["a"], paths)["a"].tolist()
    assert got == exp == [3, 2, 1, 4, 2]


def test_zero_row_partition_between_full_ones(tmp_path):
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    a = pd.DataFrame({"a": np.array([1, 2, 2], dtype="int32"), "x": [1.0, 2.0, 3.0]})
    empty = a.iloc[:0]
    b = pd.DataFrame({"a": np.array([2, 3], dtype="int32"), "x": [4.0, np.nan]})
    ds = nvt.Dataset([a, empty, b])
    wf = nvt.Workflow((["a"] >> ops.Categorify(out_path=str(tmp_path / "g")))
                      + (["x"] >> ops.FillMissing() >> ops.Normalize()))
    got = wf.fit_transform(ds).to_ddf().compute()
    full = pd.concat([a, b], ignore_index=True)
    paths = O.categorify_fit([a, b], ["a"], str(tmp_path / "c"), tie_break="stable")
    exp = O.categorify_transform(full, ["a"], paths)
    np.testing.assert_array_equal(got["a"].to_numpy(), exp["a"].to_numpy())
    assert len(got) == 5


def test_row_limit_is_reported_not_silently_wrapped():
    """nvt_dense_count_* counts in 32 bits per call: 2**32 rows must be refused loudly."""
    import ctypes as C

    from nvtabular_amd import _lib

    lib = _lib.load()
    state = torch.zeros(_lib.STATE_WORDS, dtype=torch.int64, device="cuda")
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    k = torch.zeros(16, dtype=torch.int32, device="cuda")
    o = torch.zeros(16, dtype=torch.int64, device="cuda")
    rc = lib.nvt_dense_count_i32(k.data_ptr(), None, None, C.c_uint64(1 << 32), 0, ws.data_ptr(),
                                 k.data_ptr(), o.data_ptr(), 16, state.data_ptr(), None)
    assert rc == -1 and b"2^32" in lib.nvt_last_error()


def test_single_column_abi_call_with_hot_filter_samples_inline():
    """nvt_dense_count_i32(path | NVT_PATH_HOT) without the batched call's per-column image:
    the hot-key sample is taken inside the column's pipeline and kept in the workspace."""
    import ctypes as C

    from nvtabular_amd import _lib
    from nvtabular_amd import kernels as K

    lib = _lib.load()
    rng = np.random.default_rng(3)
    n = 1_200_000
    ids = (rng.zipf(1.2, n) % 300_000).astype("int32") * 11 - 7
    keys = torch.from_numpy(ids).cuda()
    path = 1 | K.PATH_HOT
    need = C.c_uint64()
    K.check(lib.nvt_dense_count_ws_bytes(4, n, path, 0, C.byref(need)))
    ws = torch.empty(need.value, dtype=torch.uint8, device="cuda")
    out_k = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    out_c = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    state = torch.zeros(_lib.STATE_WORDS, dtype=torch.int64, device="cuda")
    K.check(lib.nvt_dense_count_i32(keys.data_ptr(), None, None, n, path, ws.data_ptr(),
                                    out_k.data_ptr(), out_c.data_ptr(), n + 1, state.data_ptr(),
                                    K.stream_ptr()))
    st = state.cpu().tolist()
    assert st[_lib.ST_OVERFLOW] == 0 and st[_lib.ST_ROWS] == n
    m = st[_lib.ST_OCCUPIED]
    got = pd.Series(out_c[:m].cpu().numpy(), index=out_k[:m].cpu().numpy()).sort_index()
    exp = pd.Series(np.ones(n, dtype="int64")).groupby(ids).sum()
    assert got.index.is_unique
    np.testing.assert_array_equal(got.index.to_numpy(), exp.index.to_numpy())
    np.testing.assert_array_equal(got.to_numpy(), exp.to_numpy())
    # the filter is refused where it does not apply
    assert lib.nvt_dense_count_ws_bytes(8, n, path, 0, C.byref(need)) == -1
    assert lib.nvt_dense_count_ws_bytes(4, n, 0 | K.PATH_HOT, 0, C.byref(need)) == -1


def test_prefix_distinct_sketch_against_exact_counts():
    """nvt_prefix_distinct (HyperLogLog of a column prefix; steers the first counting path of a fit
    without hints): within 8 % of the exact distinct count from 1 key to 200 k keys, valid rows
    exact, int32 and int64 keys, a prefix shorter than the column, an all-null column."""
    from nvtabular_amd import _lib
    from nvtabular_amd import kernels as K
    from nvtabular_amd.device import pack_bitmap_device

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    n = 262_144
    cases = []
    for card, dt in ((1, "int32"), (7, "int32"), (300, "int64"), (5000, "int32"), (60_000, "int64"), (10**9, "int32")):
        k = rng.integers(0, card, n).astype(dt) * (7919 if dt == "int32" else 1_000_003)
        cases.append((k, None, n))
    kz = (np.minimum(rng.zipf(1.1, n), 10**7) * 2654435761 % 2**31).astype("int32")
    mask = rng.random(n) >= 0.3
    cases.append((kz, mask, n))
    cases.append((kz, mask, 50_000))                      # only the first 50 000 rows
    cases.append((kz, np.zeros(n, dtype=bool), n))        # all null
    descs = (_lib.PrefixCol * len(cases))()
    keep = []
    for d, (k, m, rows) in zip(descs, cases):
        kt = torch.tensor(k, device=dev)
        vt = pack_bitmap_device(torch.tensor(m, device=dev)) if m is not None else None
        keep.append((kt, vt))
        d.keys, d.valid, d.n, d.key_bytes = kt.data_ptr(), K.ptr(vt), rows, kt.element_size()
    out = torch.empty((len(cases), 2), dtype=torch.int64, device=dev)
    K.check(_lib.load().nvt_prefix_distinct(descs, len(cases), out.data_ptr(), K.stream_ptr()), "nvt_prefix_distinct")
    got = out.cpu().numpy()
    for (k, m, rows), (est, valid_rows) in zip(cases, got):
        kk = k[:rows] if m is None else k[:rows][m[:rows]]
        exact = len(np.unique(kk))
        assert valid_rows == len(kk)
        if exact <= 10:
            assert est == exact, (est, exact)
        else:
            assert abs(est - exact) <= 0.08 * exact, (est, exact)


def test_fresh_fit_takes_the_sketch_and_matches_the_exact_presample(tmp_path, monkeypatch):
    """A fit without hints sizes its first paths from the sketch; labels equal those of the exact
    prefix count (the estimate steers paths, never results)."""
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K
    from nvtabular_amd import ops

    rng = np.random.default_rng(5)
    n = 2_200_000   # > SAMPLE_MIN_ROWS: the presample runs
    df = pd.DataFrame({"a": (np.minimum(rng.zipf(1.15, n), 300_000) * 40503 % 2**31).astype("int32"),
                       "b": rng.integers(0, 50, n).astype("int32"),
                       "c": rng.integers(0, 20_000, n).astype("int64")})
    outs = []
    for sketch in (True, False):
        monkeypatch.setattr(K, "PRESAMPLE_SKETCH", sketch)
        before = K.STATS["presampled_columns"]
        wf = nvt.Workflow(["a", "b", "c"] >> ops.Categorify(out_path=str(tmp_path / f"s{int(sketch)}")))
        outs.append(wf.fit_transform(nvt.Dataset(df)).to_ddf().compute())
        assert K.STATS["presampled_columns"] == before + 3
    for c in ("a", "b", "c"):
        np.testing.assert_array_equal(outs[0][c].to_numpy(), outs[1][c].to_numpy(), err_msg=c)
