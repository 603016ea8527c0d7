"""BASELINE.json configs[1] at FULL size (45 M rows x 26 categorical + 13 continuous, the
frame bench.py times), checked through size-independent properties -- the oracle cannot run
at this size in test time:

* every vocabulary is duplicate-free, ordered (count desc, key asc), and sum(counts) ==
  non-null rows (groupby-size bookkeeping, reference tests/unit/ops/test_categorify.py:41-96,383-388)
* encode -> decode round trip: vocab[label - 3] == key on every non-null row, nulls -> 1, no
  row falls into the OOV slot
* checksum of checksums: bincount(labels) reproduces the fit's counts exactly
* Normalize: means / stds equal an independent float64 torch reduction within 1e-6 relative
  (the north-star tolerance); transformed columns have mean ~0, std ~1
* the fit is idempotent: refitting gives the same vocabularies
"""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = int(os.environ.get("NVT_FULLSIZE_ROWS", 45_000_000))


def _valid_mask(col, n):
    if col.valid is None:
        return torch.ones(n, dtype=torch.bool, device=col.data.device)
    idx = torch.arange(n, device=col.data.device)
    return ((col.valid[idx >> 3] >> (idx & 7).to(torch.uint8)) & 1).to(torch.bool)


@pytest.mark.timeout(600)
def test_criteo_day0_full_size_properties(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    import nvtabular_amd as nvt
    from nvtabular_amd import ops

    dev = torch.device("cuda", 0)
    frame = bench.synth_criteo(ROWS, dev)
    cats = [c for c in frame.columns if c.startswith("C")]
    conts = [c for c in frame.columns if c.startswith("I")]
    wf = bench.build_workflow(cats, conts, str(tmp_path))
    ds = nvt.Dataset(frame)
    wf.fit(ds)
    out = wf.transform(frame)
    cat_op = next(n.op for n in nvt.workflow.iter_nodes(wf.output_node) if isinstance(n.op, ops.Categorify))
    norm_op = next(n.op for n in nvt.workflow.iter_nodes(wf.output_node) if isinstance(n.op, ops.Normalize))
    vocabs = {}
    for c in cats:
        final = cat_op._pending[c]
        ks, counts = cat_op.fitted_vocabulary(c)  # waits for sorts still on internal streams
        keys = ks[0]
        vocabs[c] = (keys.clone(), counts.clone())
        col = frame[c]
        ok = _valid_mask(col, ROWS)
        n_valid = int(ok.sum().item())
        # vocabulary bookkeeping
        assert int(counts.sum().item()) == n_valid, c
        assert final["null_size"] == ROWS - n_valid, c
        assert torch.unique(keys).numel() == keys.numel(), c
        dc = counts[1:] - counts[:-1]
        assert bool((dc <= 0).all()), c                                   # count descending
        tie = dc == 0
        assert bool((keys[1:][tie] > keys[:-1][tie]).all()), c            # key ascending in ties
        # encode -> decode
        lab = out[c].data
        assert lab.dtype == torch.int64
        assert bool((lab[~ok] == 1).all()), c                             # nulls
        assert int((lab == 2).sum().item()) == 0, c                       # nothing out of vocabulary
        dec = keys[(lab[ok] - 3)]
        assert bool((dec == col.data[ok]).all()), c
        # checksum of checksums
        hist = torch.bincount(lab[ok] - 3, minlength=keys.numel())
        assert bool((hist == counts).all()), c
        del lab, dec, hist, ok
    for c in conts:
        col = frame[c]
        ok = _valid_mask(col, ROWS)
        x = torch.where(ok, col.data.to(torch.float64), torch.zeros((), dtype=torch.float64, device=dev))
        mean = float(x.mean().item())
        std = float(x.std(unbiased=True).item())
        assert abs(norm_op.means[c] - mean) <= 1e-6 * abs(mean), c       # north star: 1e-6 relative
        assert abs(norm_op.stds[c] - std) <= 1e-6 * abs(std), c
        y = out[c].data
        assert y.dtype == torch.float64 and not bool(torch.isnan(y).any())
        assert abs(float(y.mean().item())) < 1e-6 and abs(float(y.std().item()) - 1.0) < 1e-6, c
        del x, y, ok
    # idempotence: a refit (now with learned cardinality hints -> other kernel paths) agrees
    wf.fit(ds)
    for c in cats:
        ks, counts = cat_op.fitted_vocabulary(c)
        assert torch.equal(ks[0], vocabs[c][0]), c
        assert torch.equal(counts, vocabs[c][1]), c


@pytest.mark.timeout(600)
def test_cfg4_target_encoding_joingroupby_high_cardinality(tmp_path):
    """BASELINE.json configs[3] scaled to one GPU: 20 M rows, 5 M-key categorical, float32
    target.  Independent check: torch scatter-adds by key (no hash tables) give per-key
    count / sum; JoinGroupby must return them per row, TargetEncoding (kfold=1) must equal
    (sum + p*ybar) / (count + p) -- target_encoding.py:360-363 -- within float32 rounding."""
    import nvtabular_amd as nvt
    from nvtabular_amd import ops
    from nvtabular_amd.device import DeviceColumn, DeviceFrame

    dev = torch.device("cuda", 0)
    n, card, p = int(os.environ.get("NVT_CFG4_ROWS", 20_000_000)), 5_000_000, 20.0
    g = torch.Generator(device=dev).manual_seed(7)
    raw = (torch.rand(n, device=dev, generator=g, dtype=torch.float64) ** 3 * card).to(torch.int64)
    key = ((raw * 2654435761) % (2**31)).to(torch.int32)        # scrambled ids, skewed frequencies
    y = torch.rand(n, device=dev, generator=g, dtype=torch.float32)
    frame = DeviceFrame({"k": DeviceColumn(key), "y": DeviceColumn(y)})
    te = ["k"] >> ops.TargetEncoding("y", kfold=1, p_smooth=p, out_path=str(tmp_path / "te"))
    jg = ["k"] >> ops.JoinGroupby(cont_cols=["y"], stats=["count", "sum"], out_path=str(tmp_path / "jg"))
    wf = nvt.Workflow(te + jg).fit(nvt.Dataset(frame))
    out = wf.transform(frame)
    cnt = torch.zeros(card, dtype=torch.float64, device=dev).index_add_(
        0, raw, torch.ones(n, dtype=torch.float64, device=dev))
    sm = torch.zeros(card, dtype=torch.float64, device=dev).index_add_(0, raw, y.to(torch.float64))
    ybar = float(y.to(torch.float64).mean().item())
    assert bool((out["k_count"].data.to(torch.float64) == cnt[raw]).all())
    assert int(out["k_count"].data.dtype == torch.int32)
    torch.testing.assert_close(out["k_y_sum"].data.to(torch.float64), sm[raw], rtol=1e-5, atol=1e-6)
    exp = ((sm[raw] + p * ybar) / (cnt[raw] + p)).to(torch.float32)
    torch.testing.assert_close(out["TE_k_y"].data, exp, rtol=2e-6, atol=1e-7)


@pytest.mark.timeout(600)
def test_cfg5_multihot_lists_categorify_hashbucket(tmp_path):
    """BASELINE.json configs[4]: multi-hot list column (power-law lengths 1..20, 1 M leaf
    cardinality, 4 M rows ~ 16 M leaves), Categorify(freq_threshold) + HashBucket(2**20).
    Properties: offsets untouched; leaves with count >= threshold decode back through the
    vocabulary, rarer ones land in the OOV slot; bincount(labels) == vocabulary counts; hash
    buckets equal the oracle's hash on a sample and stay in range."""
    import numpy as np

    import nvtabular_amd as nvt
    import oracle as O
    from nvtabular_amd import ops
    from nvtabular_amd.device import DeviceColumn, DeviceFrame

    dev = torch.device("cuda", 0)
    rows, card, thr = 4_000_000, 1_000_000, 3
    g = torch.Generator(device=dev).manual_seed(11)
    lens = (20.0 ** torch.rand(rows, device=dev, generator=g)).floor().clamp_(1, 20).to(torch.int64)
    offsets = torch.zeros(rows + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=offsets[1:])
    total = int(offsets[-1].item())
    raw = (torch.rand(total, device=dev, generator=g, dtype=torch.float64) ** 4 * card).to(torch.int64)
    leaves = ((raw * 2654435761) % (2**31)).to(torch.int32)
    frame = DeviceFrame({"tags": DeviceColumn(leaves, None, offsets),
                         "tags_h": DeviceColumn(leaves.clone(), None, offsets)})
    cat = ops.Categorify(out_path=str(tmp_path), freq_threshold=thr, defer_artifacts=True)
    wf = nvt.Workflow((["tags"] >> cat) + (["tags_h"] >> ops.HashBucket(2**20)))
    wf.fit(nvt.Dataset(frame))
    out = wf.transform(frame)
    assert torch.equal(out["tags"].offsets, offsets) and torch.equal(out["tags_h"].offsets, offsets)
    lab = out["tags"].data
    ks, counts = cat.fitted_vocabulary("tags")
    keys = ks[0]
    true_cnt = torch.zeros(card, dtype=torch.int64, device=dev).index_add_(
        0, raw, torch.ones(total, dtype=torch.int64, device=dev))
    frequent = true_cnt[raw] >= thr
    assert bool((lab[~frequent] == 2).all())                       # below the threshold -> OOV
    assert bool((keys[lab[frequent] - 3] == leaves[frequent]).all())  # decode round trip
    assert bool((torch.bincount(lab[frequent] - 3, minlength=keys.numel()) == counts).all())
    assert int(counts.min().item()) >= thr and keys.numel() == int((true_cnt >= thr).sum().item())
    hb = out["tags_h"].data
    assert hb.dtype == torch.int32 and int(hb.min().item()) >= 0 and int(hb.max().item()) < 2**20
    samp = leaves[:100_000].cpu().numpy()
    exp = (O.nvt_hash32(samp.astype(np.int64)) % np.uint32(2**20)).astype(np.int32)
    np.testing.assert_array_equal(hb[:100_000].cpu().numpy(), exp)


@pytest.mark.timeout(600)
def test_cfg3_high_cardinality_column_stays_on_partitioned_path():
    """BASELINE.json configs[2]'s worst columns (Criteo-1TB C1/C10/C20/C22: ~4e7 uniques):
    64 M rows of uniform draws from 4.8e7 ids -> ~3.5e7 distinct keys in one partition.  The
    groupby-size must come from the sort path (int32 keys) or the partitioned LDS path 3, never
    the global-atomic fallback, and equal torch.unique exactly; the encode round-trips."""
    import torch

    from nvtabular_amd import kernels as K

    dev = torch.device("cuda", 0)
    n, card = 64_000_000, 48_000_000
    g = torch.Generator(device=dev).manual_seed(7)
    keys = (torch.randint(0, card, (n,), device=dev, generator=g, dtype=torch.int64) * 2654435761 % (2**31)
            ).to(torch.int32)
    uk, uc = torch.unique(keys, return_counts=True)
    # int32 keys without weights: the sort path (key-sorted output, no re-sort needed) ...
    k, c, nulls, info = K.dense_count(keys, None, hint=40_000_000)
    assert info["path"] == K.PATH_SORT and info["sorted_by_key"], info["path"]
    assert info["distinct"] == uk.numel() > 32_000_000
    assert torch.equal(k, uk) and torch.equal(c, uc)
    # ... and the hash-partitioned path 3 (what int64 keys / weighted merges of this size take)
    job = K.DenseCountJob(keys, None, None, hint=40_000_000)
    job.path = 3
    k, c, nulls, info = K.dense_count_many([job])[0]
    assert info["path"] == 3, info["path"]
    order = torch.argsort(k)
    assert torch.equal(k[order], uk) and torch.equal(c[order], uc)
    del order, uk, uc
    # vocabulary order + encode table at this size: labels decode back to the keys
    K.vocab_sort(k, c, info["max_count"])
    tab = K.EncodeTable(k, 3, unique=True)
    lab = tab.encode(keys, None, 1, 2)
    assert int((lab < 3).sum().item()) == 0
    assert torch.equal(k[(lab - 3)], keys)


@pytest.mark.timeout(900)
def test_bench_generator_5m_rows_vs_oracle(tmp_path):
    """the largest direct HIP-vs-oracle comparison was 2 M rows.  5 M rows of the
    bench generator (26 Criteo-cardinality categoricals + 13 continuous, nulls as bitmaps): the
    partitioned counting paths (1 / 2), the onesweep vocabulary sort, the 2-choice encode cache
    and the batched moments / fill+normalize launches against the pandas restatement -- labels
    bit-exact, means / stds / normalised values within 1e-6 relative."""
    import torch

    import bench
    import nvtabular_amd as nvt
    import oracle as O
    from nvtabular_amd.node import iter_nodes

    dev = torch.device("cuda", 0)
    n = 5_000_000
    frame = bench.synth_criteo(n, dev)
    cats = [c for c in frame.columns if c.startswith("C")]
    conts = [c for c in frame.columns if c.startswith("I")]
    wf = bench.build_workflow(cats, conts, str(tmp_path / "gpu"))
    wf.fit(nvt.Dataset(frame))
    out = wf.transform(frame)
    cat_op = next(x.op for x in iter_nodes(wf.output_node) if type(x.op).__name__ == "Categorify")
    paths_used = {cat_op._cap_hints[f"{c}#0"] for c in cats}
    assert max(paths_used) > 500_000  # high-cardinality columns: the partitioned counting path
    df = bench.frame_to_oracle_pandas(frame, n)
    paths = O.categorify_fit([df], cats, str(tmp_path / "cpu"), tie_break="stable")
    exp = O.categorify_transform(df, cats, paths)
    for c in cats:
        got = out[c].data.cpu().numpy()
        assert (got == exp[c].to_numpy()).all(), c
    filled = O.fill_missing(df[conts].copy(), conts, 0)
    mom = O.custom_moments([filled], conts)
    ref = O.normalize_transform(filled, conts, mom["mean"].to_dict(), mom["std"].to_dict())
    norm_op = next(x.op for x in iter_nodes(wf.output_node) if type(x.op).__name__ == "Normalize")
    for c in conts:
        assert abs(norm_op.means[c] - float(mom["mean"][c])) <= 1e-6 * abs(float(mom["mean"][c])), c
        assert abs(norm_op.stds[c] - float(mom["std"][c])) <= 1e-6 * abs(float(mom["std"][c])), c
        g = out[c].data.cpu().numpy()
        e = ref[c].to_numpy()
        assert float(np.max(np.abs(g - e) / np.maximum(np.abs(e), 1.0))) <= 1e-6, c


@pytest.mark.timeout(300)
def test_merge_counts_sorted_at_exchange_size():
    """The owner-side merge at the size an 8-rank Criteo fit hands it (~31 M received rows in
    8 x 26 segments; here 40 M): per column the keys come out strictly ascending, the counts sum
    to what was received, and a column present in one segment only is returned unchanged."""
    from nvtabular_amd import kernels as K

    dev = torch.device("cuda", 0)
    G, ncol, n = 8, 26, 40_000_000
    g = torch.Generator(device=dev).manual_seed(3)
    per = n // (G * ncol)
    off = [s * per for s in range(G * ncol + 1)]
    n = off[-1]
    # column j draws its keys from a range of (j + 1) * 100 k ids: heavy overlap between sources
    col_of_row = (torch.arange(n, device=dev) // per) % ncol
    span = (col_of_row + 1) * 100_000
    keys = (torch.rand(n, device=dev, generator=g, dtype=torch.float64) * span).to(torch.int64) - 50_000
    cnt = torch.randint(1, 1000, (n,), device=dev, dtype=torch.int64, generator=g)
    rows = (cnt << 32) | (keys & 0xFFFFFFFF)
    out = K.merge_counts_sorted(rows, off, ncol)
    assert len(out) == ncol
    for j, (mk, mc) in enumerate(out):
        sel = col_of_row == j
        assert bool((mk[1:] > mk[:-1]).all()), j
        assert int(mc.sum().item()) == int(cnt[sel].sum().item()), j
        assert mk.numel() == int(torch.unique(keys[sel]).numel()), j
    # exactness on one column against a scatter-add
    j = 5
    sel = col_of_row == j
    kj = keys[sel] + 50_000
    ref = torch.zeros(int(kj.max().item()) + 1, dtype=torch.int64, device=dev).index_add_(0, kj, cnt[sel])
    mk, mc = out[j]
    assert torch.equal(ref[(mk.to(torch.int64) + 50_000)], mc)
