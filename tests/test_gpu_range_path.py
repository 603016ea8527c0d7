"""Range path (NVT_PATH_RANGE, csrc/nvt_range_count.hip) + one-pass vocabulary ordering
(csrc/nvt_sort.hip cls_scatter): exact counts, key-ordered output, overflow fallback, and the
(count desc, key asc) order / encode table built from it -- against numpy / the oracle.
Reference semantics: categorify.py:955-1051 (groupby-size), :1300,1316 (order), :1558-1807."""
import numpy as np
import pandas as pd
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _count_forced(keys, valid, path, hint):
    from nvtabular_amd import kernels as K

    job = K.DenseCountJob(keys, valid, None, hint=hint)
    job.path = path
    return K.dense_count_many([job])[0]


def _zipf_keys(rng, n, card, s=1.15, scramble=True):
    u = rng.random(n)
    x = np.floor(((card ** (1.0 - s) - 1.0) * u + 1.0) ** (1.0 / (1.0 - s))).clip(1, card).astype(np.int64)
    if scramble:
        x = (x * 2654435761 + 12345) % (2**31)
    return x.astype(np.int32)


@pytest.mark.parametrize("n,card,hint", [
    (300_000, 40_000, 30_000),        # 256 buckets, few keys per bucket
    (2_000_000, 3_000_000, 1_500_000),  # 512 buckets
    (3_000_000, 40_000_000, 6_000_000),  # 1024 buckets, mostly singletons
    (1003, 500, 20_000),              # fewer rows than workgroups, ragged tail
])
@pytest.mark.parametrize("nulls", [False, True])
def test_range_path_counts_exact_and_key_ordered(n, card, hint, nulls):
    from nvtabular_amd import kernels as K
    from nvtabular_amd.device import pack_bitmap

    rng = np.random.default_rng(n + card)
    ids = _zipf_keys(rng, n, card)
    ids[7] = np.iinfo(np.int32).min  # the sentinel key is a legal value
    ids[11] = np.iinfo(np.int32).max
    ids[13] = -5
    mask = rng.random(n) < 0.1 if nulls else np.zeros(n, dtype=bool)
    mask[7] = False
    keys = torch.from_numpy(ids).cuda()
    valid = torch.from_numpy(pack_bitmap(~mask)).cuda() if nulls else None
    k, c, nn, info = _count_forced(keys, valid, K.PATH_RANGE, hint)
    assert info["path"] == K.PATH_RANGE and info["sorted_by_key"]
    exp = pd.Series(np.ones(n, dtype=np.int64)[~mask]).groupby(ids[~mask]).sum()
    hk, hc = k.cpu().numpy(), c.cpu().numpy()
    assert nn == int(mask.sum()) and info["rows"] == n
    np.testing.assert_array_equal(hk, exp.index.to_numpy())        # key order, sentinel first
    np.testing.assert_array_equal(hc, exp.to_numpy())
    assert info["max_count"] == int(hc.max())
    hist = info["cls_hist"].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    np.testing.assert_array_equal(hist, np.bincount(np.minimum(hc, 255), minlength=256))
    assert info["n_big"] == int((hc >= 255).sum())


def test_range_path_falls_back_when_keys_are_not_spread():
    """Dense ids plus one far outlier: the sampled range is a poor partition (every key in one
    bucket); the overflow bit sends the column to a hash path and the result stays exact."""
    from nvtabular_amd import kernels as K

    rng = np.random.default_rng(3)
    n = 1_500_000
    ids = rng.integers(0, 900_000, n).astype(np.int32)
    ids[::1000] = 2**31 - 7
    keys = torch.from_numpy(ids).cuda()
    k, c, nn, info = _count_forced(keys, None, K.PATH_RANGE, 900_000)
    assert info["path"] != K.PATH_RANGE and info["range_failed"]
    got = pd.Series(c.cpu().numpy(), index=k.cpu().numpy()).sort_index()
    exp = pd.Series(np.ones(n, dtype=np.int64)).groupby(ids).sum()
    np.testing.assert_array_equal(got.index.to_numpy(), exp.index.to_numpy())
    np.testing.assert_array_equal(got.to_numpy(), exp.to_numpy())


def test_range_path_dense_ids_stay_on_the_range_path():
    """Dense ids 0..N-1 (the usual already-encoded column): direct-address like, no fallback."""
    from nvtabular_amd import kernels as K

    rng = np.random.default_rng(4)
    n = 2_000_000
    ids = rng.integers(0, 1_200_000, n).astype(np.int32)
    k, c, nn, info = _count_forced(torch.from_numpy(ids).cuda(), None, K.PATH_RANGE, 1_000_000)
    assert info["path"] == K.PATH_RANGE
    exp = pd.Series(np.ones(n, dtype=np.int64)).groupby(ids).sum()
    np.testing.assert_array_equal(k.cpu().numpy(), exp.index.to_numpy())
    np.testing.assert_array_equal(c.cpu().numpy(), exp.to_numpy())


@pytest.mark.parametrize("n,card", [(1_000_000, 400_000), (3_000_000, 5_000_000)])
def test_workflow_on_the_range_path_vs_oracle(tmp_path, n, card):
    """Categorify fit + transform where the column takes the range path and the one-pass
    ordering: labels bit-exact against the oracle (count desc, value asc), unique.*.parquet
    identical; a second fit (steady state: hints known) gives the same labels."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops
    from nvtabular_amd.device import DeviceColumn, DeviceFrame, pack_bitmap

    rng = np.random.default_rng(card)
    ids = _zipf_keys(rng, n, card, s=1.1)
    mask = rng.random(n) < 0.03
    frame = DeviceFrame({"c": DeviceColumn(torch.from_numpy(ids).cuda(),
                                           torch.from_numpy(pack_bitmap(~mask)).cuda())})
    op = ops.Categorify(out_path=str(tmp_path / "g"))
    wf = nvt.Workflow(["c"] >> op)
    wf.fit(nvt.Dataset(frame))
    got = wf.transform(frame)["c"].data.cpu().numpy()
    assert op._last_paths["c#0"] == 9
    vals = ids.astype("float64")
    vals[mask] = np.nan
    df = pd.DataFrame({"c": vals})
    paths = O.categorify_fit([df], ["c"], str(tmp_path / "c"), tie_break="stable")
    exp = O.categorify_transform(df, ["c"], paths)["c"].to_numpy()
    np.testing.assert_array_equal(got, exp)
    a = pd.read_parquet(tmp_path / "g" / "categories" / "unique.c.parquet")
    b = pd.read_parquet(paths["c"])
    np.testing.assert_array_equal(a["c"].to_numpy().astype(np.int64), b["c"].to_numpy().astype(np.int64))
    np.testing.assert_array_equal(a["c_size"].to_numpy(), b["c_size"].to_numpy())
    wf.fit(nvt.Dataset(frame))
    again = wf.transform(frame)["c"].data.cpu().numpy()
    np.testing.assert_array_equal(again, exp)


@pytest.mark.parametrize("n,card", [(1003, 400), (300_000, 10**9), (2_500_000, 3_000_000)])
@pytest.mark.parametrize("nulls", [False, True])
def test_sort_path_counts_exact_and_key_ordered(n, card, nulls):
    """NVT_PATH_SORT (csrc/nvt_sort_count.hip): radix sort of the rows + run lengths -- exact
    counts in key order whatever the key distribution, class histogram for the ordering pass."""
    from nvtabular_amd import kernels as K
    from nvtabular_amd.device import pack_bitmap

    rng = np.random.default_rng(n)
    ids = rng.integers(-card, card, n).astype(np.int64).clip(-2**31, 2**31 - 1).astype(np.int32)
    ids[3] = np.iinfo(np.int32).min
    ids[5] = np.iinfo(np.int32).max
    ids[100:400] = 77                      # one key with a count >= 255
    mask = rng.random(n) < 0.2 if nulls else np.zeros(n, dtype=bool)
    keys = torch.from_numpy(ids).cuda()
    valid = torch.from_numpy(pack_bitmap(~mask)).cuda() if nulls else None
    k, c, nn, info = _count_forced(keys, valid, K.PATH_SORT, 1000)  # (too small a hint: relaunch)
    assert info["path"] == K.PATH_SORT and info["sorted_by_key"]
    exp = pd.Series(np.ones(n, dtype=np.int64)[~mask]).groupby(ids[~mask]).sum()
    np.testing.assert_array_equal(k.cpu().numpy(), exp.index.to_numpy())
    np.testing.assert_array_equal(c.cpu().numpy(), exp.to_numpy())
    assert nn == int(mask.sum()) and info["max_count"] == int(exp.max())
    hist = info["cls_hist"].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    np.testing.assert_array_equal(hist, np.bincount(np.minimum(exp.to_numpy(), 255), minlength=256))
    assert info["n_big"] == int((exp.to_numpy() >= 255).sum())


def test_workflow_on_the_sort_path_vs_oracle(tmp_path):
    """A column beyond the range path (forced with a small NVT_RANGE_MAX): sort path -> one-pass
    ordering -> hashed encode table; labels and unique.*.parquet equal the oracle's."""
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K, ops
    from nvtabular_amd.device import DeviceColumn, DeviceFrame

    rng = np.random.default_rng(8)
    n = 3_000_000
    ids = rng.integers(0, 4_000_000, n).astype(np.int32)
    old = K.PATH_RANGE_MAX_DISTINCT
    K.PATH_RANGE_MAX_DISTINCT = 100_000
    try:
        frame = DeviceFrame({"c": DeviceColumn(torch.from_numpy(ids).cuda())})
        op = ops.Categorify(out_path=str(tmp_path / "g"))
        wf = nvt.Workflow(["c"] >> op)
        wf.fit(nvt.Dataset(frame))
        got = wf.transform(frame)["c"].data.cpu().numpy()
        assert op._last_paths["c#0"] == K.PATH_SORT
    finally:
        K.PATH_RANGE_MAX_DISTINCT = old
    df = pd.DataFrame({"c": ids})
    paths = O.categorify_fit([df], ["c"], str(tmp_path / "c"), tie_break="stable")
    exp = O.categorify_transform(df, ["c"], paths)["c"].to_numpy()
    np.testing.assert_array_equal(got, exp)
    a = pd.read_parquet(tmp_path / "g" / "categories" / "unique.c.parquet")
    b = pd.read_parquet(paths["c"])
    np.testing.assert_array_equal(a["c"].to_numpy().astype(np.int64), b["c"].to_numpy().astype(np.int64))
    np.testing.assert_array_equal(a["c_size"].to_numpy(), b["c_size"].to_numpy())


def test_flat_table_falls_back_to_a_hashed_table_for_clustered_keys(tmp_path):
    """Two dense clusters of keys 2^30 apart: in a monotone table every cluster is one long probe
    run (flat_build_kernel reports the displacement); Categorify then builds a hashed table.  Labels
    stay exact either way; the spread-out case keeps the flat table."""
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K, ops
    from nvtabular_amd.device import DeviceColumn, DeviceFrame

    rng = np.random.default_rng(9)
    n = 2_400_000  # (>= 2 M rows: the cold fit estimates the distinct count from a prefix)
    old = K.PATH_RANGE_MAX_DISTINCT
    K.PATH_RANGE_MAX_DISTINCT = 50_000
    try:
        for clustered in (True, False):
            if clustered:
                ids = rng.integers(0, 300_000, n).astype(np.int64)
                ids[n // 2:] += 2**30
            else:
                ids = rng.integers(0, 2**31 - 1, n).astype(np.int64)
            ids = ids.astype(np.int32)
            frame = DeviceFrame({"c": DeviceColumn(torch.from_numpy(ids).cuda())})
            op = ops.Categorify(out_path=str(tmp_path / f"g{int(clustered)}"))
            wf = nvt.Workflow(["c"] >> op)
            wf.fit(nvt.Dataset(frame))
            got = wf.transform(frame)["c"].data.cpu().numpy()
            assert op._last_paths["c#0"] == K.PATH_SORT
            tab = op._encoders["c"].table
            assert (tab.flat_slots == 0) == clustered, (clustered, tab.flat_slots)
            df = pd.DataFrame({"c": ids})
            paths = O.categorify_fit([df], ["c"], str(tmp_path / f"c{int(clustered)}"), tie_break="stable")
            exp = O.categorify_transform(df, ["c"], paths)["c"].to_numpy()
            np.testing.assert_array_equal(got, exp)
    finally:
        K.PATH_RANGE_MAX_DISTINCT = old


def test_dense_ids_take_the_range_path_with_splitters(tmp_path):
    """Dense frequency-ordered ids (id = rank, power law): the linear range map overflows, the
    sort path delivers the exact list, kernels.range_splitters turns it into a piecewise map and
    the NEXT fit counts the column on the range path (NVT_PATH_PIECES) -- count, ordering, the
    dumped table addressed through the piecewise map and the encode all agree with the oracle."""
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K, ops
    from nvtabular_amd.device import DeviceColumn, DeviceFrame

    rng = np.random.default_rng(12)
    n, card, s = 3_000_000, 2_000_000, 1.1
    u = rng.random(n)
    ids = np.floor(((card ** (1.0 - s) - 1.0) * u + 1.0) ** (1.0 / (1.0 - s))).clip(1, card).astype(np.int32)
    ids[5] = np.iinfo(np.int32).min   # the sentinel key leads the sorted list
    frame = DeviceFrame({"c": DeviceColumn(torch.from_numpy(ids).cuda())})
    op = ops.Categorify(out_path=str(tmp_path / "g"))
    wf = nvt.Workflow(["c"] >> op)
    wf.fit(nvt.Dataset(frame))
    first = op._last_paths["c#0"]
    got1 = wf.transform(frame)["c"].data.cpu().numpy()
    wf.fit(nvt.Dataset(frame))
    got2 = wf.transform(frame)["c"].data.cpu().numpy()
    df = pd.DataFrame({"c": ids})
    paths = O.categorify_fit([df], ["c"], str(tmp_path / "c"), tie_break="stable")
    exp = O.categorify_transform(df, ["c"], paths)["c"].to_numpy()
    np.testing.assert_array_equal(got1, exp)
    np.testing.assert_array_equal(got2, exp)
    if first == K.PATH_SORT:   # the linear map overflowed on the first fit (expected for these ids)
        assert "c#0" in op._range_pieces and op._last_paths["c#0"] == K.PATH_RANGE
        assert "c#0" not in op._no_range


def test_range_splitters_balance_rows_and_keys():
    from nvtabular_amd import kernels as K

    rng = np.random.default_rng(2)
    keys = np.unique(np.concatenate([np.arange(1, 200_000), rng.integers(200_000, 2**31 - 1, 300_000)])).astype(np.int32)
    counts = np.maximum(1, (3e6 / np.arange(1, keys.size + 1) ** 1.1)).astype(np.int64)
    sp = K.range_splitters(torch.from_numpy(keys).cuda(), torch.from_numpy(counts).cuda())
    u = sp.cpu().numpy().view(np.uint32).astype(np.int64)
    assert u.size == 65 and (np.diff(u) > 0).all()
    assert u[0] == int(keys[0]) + 2**31 and u[-1] == int(keys[-1]) + 2**31 + 1
    piece = np.searchsorted(u, keys.astype(np.int64) + 2**31, side="right") - 1
    distinct = np.bincount(piece, minlength=64)
    assert distinct.max() <= 2.0 * keys.size / 64      # <= 1.5 x by construction, 2 x with slack
