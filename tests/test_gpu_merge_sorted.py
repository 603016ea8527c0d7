"""Merge of key-sorted partial results (csrc/nvt_merge.hip) and the multi-partition Categorify fit
built on it -- against numpy / pandas and the oracle.
Reference semantics: _mid_level_groupby (categorify.py:1054-1070) inside the tree of
categorify.py:1423-1478: concat of the partial (key, size) frames + groupby-sum."""
import numpy as np
import pandas as pd
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TILE = 1024 * 5   # kMgTile of nvt_merge.hip


def _np_merge(ka, ca, kb, cb):
    s = pd.concat([pd.Series(ca, index=ka), pd.Series(cb, index=kb)]).groupby(level=0).sum().sort_index()
    return s.index.to_numpy().astype(np.int32), s.to_numpy().astype(np.int64)


def _rand_list(rng, n, lo=-2**31, hi=2**31 - 1):
    k = np.unique(rng.integers(lo, hi, int(n * 1.2) + 8).astype(np.int32))[:n]
    return k, rng.integers(1, 1000, k.size).astype(np.int64)


def _dev(*arrs):
    return tuple(torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrs)


def _check_pairs(pairs_np, want_src=False):
    from nvtabular_amd import kernels as K

    pairs = [(_dev(ka, ca), _dev(kb, cb)) for ka, ca, kb, cb in pairs_np]
    got = K.merge_sorted_pairs(pairs, want_src=want_src)
    for (ka, ca, kb, cb), res in zip(pairs_np, got):
        ek, ec = _np_merge(ka, ca, kb, cb)
        np.testing.assert_array_equal(res[0].cpu().numpy(), ek)
        np.testing.assert_array_equal(res[1].cpu().numpy(), ec)
        if want_src:
            sa, sb = res[2].cpu().numpy(), res[3].cpu().numpy()
            assert ((sa >= 0) | (sb >= 0)).all()
            np.testing.assert_array_equal(ka[sa[sa >= 0]], ek[sa >= 0])
            np.testing.assert_array_equal(kb[sb[sb >= 0]], ek[sb >= 0])
            assert (sa >= 0).sum() == ka.size and (sb >= 0).sum() == kb.size
            ca0, cb0 = np.append(ca, 0), np.append(cb, 0)   # (an empty list: nothing to index)
            rec = np.where(sa >= 0, ca0[np.maximum(sa, 0)], 0) + np.where(sb >= 0, cb0[np.maximum(sb, 0)], 0)
            np.testing.assert_array_equal(rec, ec)


@pytest.mark.parametrize("want_src", [False, True])
def test_merge_pairs_edge_shapes(want_src):
    rng = np.random.default_rng(3)
    e = (np.empty(0, np.int32), np.empty(0, np.int64))
    a1, b1 = _rand_list(rng, 1000), _rand_list(rng, 1500)
    dense = (np.arange(-3000, 9000, dtype=np.int32), rng.integers(1, 9, 12000).astype(np.int64))
    ext = (np.array([-2**31, -1, 0, 2**31 - 1], np.int32), np.array([5, 6, 7, 8], np.int64))
    # equal keys that straddle tile and thread boundaries: identical lists of several tiles
    same = _rand_list(rng, 3 * TILE + 17, -50_000, 50_000)
    # B entirely below / above A, interleaved odd / even
    lo = (np.arange(-9000, -1000, dtype=np.int32), np.ones(8000, np.int64))
    odd = (np.arange(1, 2 * TILE * 2, 2, dtype=np.int32), np.full(2 * TILE, 3, np.int64))
    even = (np.arange(0, 2 * TILE * 2, 2, dtype=np.int32), np.full(2 * TILE, 4, np.int64))
    pairs = [a1 + b1, a1 + a1, e + b1, a1 + e, e + e, dense + a1, ext + ext, ext + dense,
             same + same, lo + dense, dense + lo, odd + even, even + odd,
             (same[0][::2], same[1][::2]) + same, same + (same[0][1::3], same[1][1::3])]
    _check_pairs(pairs, want_src)


@pytest.mark.parametrize("na,nb", [(TILE, TILE), (TILE - 1, 1), (1, TILE), (5 * TILE + 3, 2 * TILE - 7),
                                   (400_000, 3_000_000)])
def test_merge_pairs_sizes_around_tiles(na, nb):
    rng = np.random.default_rng(na + nb)
    a, b = _rand_list(rng, na, -4_000_000, 4_000_000), _rand_list(rng, nb, -4_000_000, 4_000_000)
    _check_pairs([a + b, b + a])


def test_merge_many_columns_one_call_and_tree():
    from nvtabular_amd import kernels as K

    rng = np.random.default_rng(11)
    # more pairs than one launch batches (30), wildly different sizes
    pairs = []
    for i in range(37):
        a = _rand_list(rng, int(rng.integers(0, 20_000)), -100_000, 100_000)
        b = _rand_list(rng, int(rng.integers(0, 20_000)), -100_000, 100_000)
        pairs.append(a + b)
    _check_pairs(pairs)
    # tree: 9 lists per column (8 partitions + the table so far), 3 columns
    cols, exp = [], []
    for j in range(3):
        lists = [_rand_list(rng, int(rng.integers(1, 30_000)), -200_000, 200_000) for _ in range(8)]
        lists.append(_rand_list(rng, 150_000, -200_000, 200_000))
        s = pd.concat([pd.Series(c, index=k) for k, c in lists]).groupby(level=0).sum().sort_index()
        exp.append(s)
        cols.append([_dev(k, c) for k, c in lists])
    cols.append([])  # a column without lists
    got = K.merge_sorted_tree(cols)
    for (k, c), s in zip(got[:3], exp):
        np.testing.assert_array_equal(k.cpu().numpy(), s.index.to_numpy().astype(np.int32))
        np.testing.assert_array_equal(c.cpu().numpy(), s.to_numpy())
    assert got[3][0].numel() == 0


def test_merge_payload_ops():
    from nvtabular_amd import kernels as K

    rng = np.random.default_rng(5)
    (ka, ca), (kb, cb) = _rand_list(rng, 7000, 0, 20_000), _rand_list(rng, 9000, 0, 20_000)
    (k, c, sa, sb), = K.merge_sorted_pairs([(_dev(ka, ca), _dev(kb, cb))], want_src=True)
    w = 3
    pa, pb = rng.normal(size=(ka.size, w)), rng.normal(size=(kb.size, w))
    pa[5, 1] = np.nan   # "no value yet" of min / max payloads
    hsa, hsb = sa.cpu().numpy(), sb.cpu().numpy()
    A = np.where((hsa >= 0)[:, None], pa[np.maximum(hsa, 0)], np.nan)
    B = np.where((hsb >= 0)[:, None], pb[np.maximum(hsb, 0)], np.nan)
    da, db = _dev(pa.reshape(-1), pb.reshape(-1))
    got = {op: K.merge_payload(sa, sb, da, db, op, w).cpu().numpy().reshape(-1, w) for op in ("add", "min", "max")}
    np.testing.assert_allclose(got["add"], np.where(np.isnan(A), 0, A) + np.where(np.isnan(B), 0, B)
                               + np.where(np.isnan(A) & np.isnan(B), np.nan, 0), equal_nan=True)
    np.testing.assert_array_equal(got["min"], np.fmin(A, B))
    np.testing.assert_array_equal(got["max"], np.fmax(A, B))
    ia, ib = _dev(ca, cb)
    np.testing.assert_array_equal(K.merge_payload(sa, sb, ia, ib, "add", 1).cpu().numpy(), c.cpu().numpy())


def _zipf_keys(rng, n, card, s=1.15):
    u = rng.random(n)
    x = np.floor(((card ** (1.0 - s) - 1.0) * u + 1.0) ** (1.0 / (1.0 - s))).clip(1, card).astype(np.int64)
    return ((x * 2654435761 + 12345) % (2**31)).astype(np.int32)


@pytest.mark.parametrize("nparts", [3, 9])
def test_categorify_multi_partition_keeps_the_sorted_paths(tmp_path, nparts, monkeypatch):
    """Range-path column, sort-path column and an LDS-resident column over several partitions
    (9 = one fan-in-8 merge inside the fit + the rest at fit_end): labels bit-exact vs the oracle,
    and the big columns' vocabularies are still finalised from a key-sorted list."""
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K
    from nvtabular_amd import ops

    # (400 k-row partitions: the sort path starts where the range path ends, moved down here)
    monkeypatch.setattr(K, "PATH_RANGE_MAX_DISTINCT", 300_000)
    rng = np.random.default_rng(17 + nparts)
    n = 400_000
    parts = []
    for p in range(nparts):
        df = pd.DataFrame({
            "r": _zipf_keys(rng, n, 300_000),                                  # range path
            "s": rng.integers(-2**31, 2**31 - 1, n).astype(np.int32),          # ~all distinct: sort path
            "t": rng.integers(0, 700, n).astype(np.int32),                      # LDS-resident
        })
        parts.append(df)
    cat = ops.Categorify(out_path=str(tmp_path / "g"), defer_artifacts=True)
    cat._cap_hints.update({"r#0": 200_000, "s#0": 400_000, "t#0": 700})
    wf = nvt.Workflow(["r", "s", "t"] >> cat)
    wf.fit(nvt.Dataset(parts))
    assert cat._last_paths["r#0"] == K.PATH_RANGE and cat._last_paths["s#0"] == K.PATH_SORT
    got = wf.transform(nvt.Dataset(parts)).to_ddf().compute()
    for name in ("r", "s"):
        tab = cat._encoders[name].table
        assert tab.flat_slots > 0 or tab.range_aux is not None   # ordered from a key-sorted list
    paths = O.categorify_fit(parts, ["r", "s", "t"], str(tmp_path / "c"), tie_break="stable")
    exp = O.categorify_transform(pd.concat(parts, ignore_index=True), ["r", "s", "t"], paths)
    for c in ("r", "s", "t"):
        np.testing.assert_array_equal(got[c].to_numpy(), exp[c].to_numpy())
    cat.flush_artifacts()
    for c in ("r", "s", "t"):
        a = pd.read_parquet(tmp_path / "g" / "categories" / f"unique.{c}.parquet")
        b = pd.read_parquet(paths[c])
        np.testing.assert_array_equal(a[c].to_numpy(), b[c].to_numpy())
        np.testing.assert_array_equal(a[f"{c}_size"].to_numpy(), b[f"{c}_size"].to_numpy())


def test_label_shard_orders_the_union_shard_by_shard():
    """nvt_vocab_label_shard: three key-range shards of one key-sorted (key, count) list are
    labelled independently (class bases of the union + the entries of each class on the shards in
    front); together with an exact sort of the entries with count >= 255 the labels equal the
    positions of numpy's "count descending, key ascending" order of the whole list."""
    from nvtabular_amd import kernels as K

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(8)
    n = 700_000
    keys = np.sort(rng.choice(2**31 - 1, n, replace=False).astype(np.int64) - 2**30).astype(np.int32)
    keys[0] = np.iinfo(np.int32).min    # the sentinel key is an entry like any other here
    counts = np.minimum(rng.zipf(1.3, n), 5_000_000).astype(np.int64)   # most 1 .. 3, some >= 255
    order = np.lexsort((keys, -counts))
    expect = np.empty(n, dtype=np.int64)
    expect[order] = np.arange(n)
    cuts = [0, 250_000, 250_001, n]     # (a one-entry shard in the middle)
    cls = np.minimum(counts, 255)
    hists = np.stack([np.bincount(cls[a:b], minlength=256) for a, b in zip(cuts[:-1], cuts[1:])])
    H = hists.sum(0)
    labels = np.full(n, -2, dtype=np.int64)
    big_rows = []
    for r, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        P = hists[:r].sum(0)
        base = np.zeros(256, dtype=np.int64)      # base(c), c = 1 .. 254
        for c in range(1, 255):
            base[c] = H[255] + H[c + 1:255].sum() + P[c]
        diff = np.zeros(256, dtype=np.int64)
        diff[255] = base[254]
        diff[2:255] = base[1:254] - base[2:255]
        diff = (diff & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        k = torch.tensor(keys[a:b], device=dev)
        c = torch.tensor(counts[a:b], device=dev)
        lab = torch.empty(b - a, dtype=torch.int32, device=dev)
        bk, bc, bs = K.label_shard(k, c, torch.tensor(diff, device=dev), int(hists[r, 255]), lab)
        lab = lab.cpu().numpy().astype(np.int64)
        small = counts[a:b] < 255
        np.testing.assert_array_equal(lab[small], expect[a:b][small])
        assert (lab[~small] == -1).all()
        src = bs.cpu().numpy()
        np.testing.assert_array_equal(src, np.nonzero(~small)[0])           # compacted in key order
        np.testing.assert_array_equal(bk.cpu().numpy(), keys[a:b][src])
        np.testing.assert_array_equal(bc.cpu().numpy(), counts[a:b][src])
        labels[a:b] = lab
        big_rows.append(a + src)
    big = np.concatenate(big_rows)                                          # key order (shards in rank order)
    pos = np.argsort(-counts[big], kind="stable")                            # count descending, key ascending
    labels[big[pos]] = np.arange(len(big))
    np.testing.assert_array_equal(labels, expect)


def test_vocabulary_and_table_from_labels(tmp_path):
    """nvt_vocab_col.src_labels: vocabulary + flat table of a key-sorted list whose entries carry
    their position in the vocabulary order == the ordering pass on the same list (ordered keys,
    counts, and the labels the table answers with)."""
    from nvtabular_amd import _lib
    from nvtabular_amd import kernels as K

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(9)
    n = 300_000
    keys = np.sort(rng.choice(2**31 - 1, n, replace=False).astype(np.int64) - 2**30).astype(np.int32)
    keys[0] = np.iinfo(np.int32).min
    counts = np.minimum(rng.zipf(1.4, n), 10**7).astype(np.int64)
    order = np.lexsort((keys, -counts))
    pos = np.empty(n, dtype=np.int32)
    pos[order] = np.arange(n, dtype=np.int32)
    k, c = torch.tensor(keys, device=dev), torch.tensor(counts, device=dev)
    res = []
    for labels in (None, torch.tensor(pos, device=dev)):
        descs = (_lib.VocabCol * 1)()
        ok, oc = torch.empty_like(k), torch.empty_like(c)
        tab = K.EncodeTable(ok, 3, unique=True, defer_build=True, range_table=None, flat=True)
        tab.fill_vocab_desc(descs[0], oc, int(counts.max()), src=(k, c, K.class_hist(c), int((counts >= 255).sum()), labels))
        K.check(_lib.load().nvt_vocab_finalize_many(descs, 1, K.stream_ptr()), "nvt_vocab_finalize_many")
        tab.wait_ready()
        probe = torch.tensor(np.concatenate([keys[::7], np.array([5, -7, 2**31 - 1], dtype=np.int32)]), device=dev)
        lab = tab.encode(probe, None, 1, 2)      # null label 1, out-of-vocabulary label 2
        res.append((ok.cpu().numpy(), oc.cpu().numpy(), lab.cpu().numpy()))
    np.testing.assert_array_equal(res[0][0], keys[order])
    np.testing.assert_array_equal(res[1][0], res[0][0])
    np.testing.assert_array_equal(res[1][1], res[0][1])
    np.testing.assert_array_equal(res[1][2], res[0][2])
    exp_lab = np.concatenate([3 + pos[::7].astype(np.int64), np.array([2, 2, 2])])
    hit = np.isin(np.array([5, -7, 2**31 - 1], dtype=np.int32), keys)
    exp_lab[-3:][hit] = res[0][2][-3:][hit]
    np.testing.assert_array_equal(res[1][2], exp_lab)
