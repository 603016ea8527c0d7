"""Lookup images (include/nvt_hip.h "Lookup images", nvt_flat_lookup_image / nvt_image_pack /
nvt_te_image): ONE probe and ONE packed per-group record per row for every operator fitted on a
key column -- JoinGroupby.transform (join_groupby.py:198-217) and TargetEncoding.transform
(target_encoding.py:341-371) on the same key.  Bit-for-bit equal to the per-operator kernels
(nvt_flat_lookup_gather / nvt_flat_lookup_te), which stay the reference implementation of the
engine, and to the oracle end to end."""
import ctypes as C

import numpy as np
import pandas as pd
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


def _keys(rng, n, card):
    ids = np.unique(rng.integers(-(2**31) + 1, 2**31 - 1, card, dtype=np.int64).astype(np.int32))
    return ids, ids[(rng.random(n) ** 2 * ids.size).astype(np.int64)]


@pytest.mark.parametrize("kfold", [0, 3, 5])
@pytest.mark.parametrize("out_dt", [torch.float32, torch.float64])
def test_image_kernels_equal_the_per_operator_kernels(kfold, out_dt):
    from nvtabular_amd import kernels as K

    dev = torch.device("cuda")
    rng = np.random.default_rng(3)
    ids, rows = _keys(rng, 200_003, 30_000)
    g = ids.size
    unseen = rng.integers(-(2**31) + 1, 2**31 - 1, 500, dtype=np.int64).astype(np.int32)
    rows = np.concatenate([rows, unseen])
    n = rows.size
    valid_np = rng.random(n) > 0.03
    index = K.FlatIndex(torch.from_numpy(ids).to(dev))
    assert index.ok()
    key = torch.from_numpy(rows).to(dev)
    from nvtabular_amd.device import pack_bitmap_device

    valid = pack_bitmap_device(torch.from_numpy(valid_np).to(dev))
    # JoinGroupby-like statistics: count (int32 out), sum (f32 out), mean (f32), a float64 one
    cnt = torch.from_numpy(rng.integers(1, 1000, g).astype(np.float64)).to(dev)
    sm = torch.from_numpy(rng.normal(size=g) * 1e3).to(dev)
    mean = sm / cnt
    wide = torch.from_numpy(rng.normal(size=g)).to(dev)
    plan = [("c", cnt, torch.int32, 0.0), ("s", sm, torch.float32, float("nan")),
            ("m", mean, torch.float32, float("nan")), ("w", wide, torch.float64, float("nan"))]
    # TargetEncoding records {sum, count, (sum_f, count_f) ...}
    kf = max(kfold, 1)
    fcnt = rng.integers(0, 5, (g, kf)).astype(np.float64)
    fsum = rng.normal(size=(g, kf)) * fcnt
    if kfold:
        rec = np.concatenate([fsum.sum(1, keepdims=True), fcnt.sum(1, keepdims=True),
                              np.stack([fsum, fcnt], axis=2).reshape(g, -1)], axis=1)
    else:
        rec = np.concatenate([fsum.sum(1, keepdims=True), fcnt.sum(1, keepdims=True) + 1], axis=1)
    rec_t = torch.from_numpy(np.ascontiguousarray(rec)).to(dev)
    # the same numbers as the fit keeps them: totals [g] + dense fold statistics [g * kfold]
    tot_c = torch.from_numpy(rec[:, 1].astype(np.int64)).to(dev)
    tot_s = torch.from_numpy(np.ascontiguousarray(rec[:, 0])).to(dev)
    fold_c = torch.from_numpy(fcnt.astype(np.int64).reshape(-1)).to(dev) if kfold else None
    fold_s = torch.from_numpy(np.ascontiguousarray(fsum.reshape(-1))).to(dev) if kfold else None
    fold = torch.from_numpy(rng.integers(0, kf, n).astype(np.uint8)).to(dev) if kfold else None
    p, ym = 20.0, 0.37

    owner = object()
    te_size = 8 if out_dt == torch.float64 else 4
    te_cons = K.LookupConsumer(
        owner, "te", (kfold + 1) * te_size, [("te", out_dt, 0, bool(kfold), ym)],
        lambda image, stride, off, groups: K.te_image(image, stride, off, tot_c, tot_s, fold_c, fold_s, kfold,
                                                     groups, p, ym, out_dt),
        (lambda m, d: fold) if kfold else None, groups=g)
    order = sorted(range(4), key=lambda i: 0 if plan[i][2] in (torch.float64, torch.int64) else 1)
    rel, at = {}, 0
    for i in order:
        rel[i] = at
        at += 8 if plan[i][2] in (torch.float64, torch.int64) else 4
    jg_cons = K.LookupConsumer(
        owner, "jg", at, [(plan[i][0], plan[i][2], rel[i], False, plan[i][3]) for i in range(4)],
        lambda image, stride, off, groups: K.image_pack(
            image, stride, [(plan[i][1], plan[i][2], off + rel[i]) for i in range(4)], groups), groups=g)
    index.attach(te_cons)
    index.attach(jg_cons)
    before = K.STATS.get("image_lookups", 0)
    with K.pass_memo():
        got_jg, flag = index.image_lookup(jg_cons, [key], [valid])     # triggers ONE launch for both
        got_te, _ = index.image_lookup(te_cons, [key], [valid], fold=fold)
    assert K.STATS["image_lookups"] == before + 1
    # reference: the per-operator kernels of the engine
    records = torch.stack([c[1] for c in plan], dim=1).contiguous()
    exp_jg, flag2 = index.gather([key], [valid], records, [c[2] for c in plan], [c[3] for c in plan])
    exp_te = index.te([key], [valid], fold, kf if kfold else 1, rec_t, p, ym, out_dt)
    for (name, _, _, _), e in zip(plan, exp_jg):
        a, b = got_jg[name].cpu().numpy(), e.cpu().numpy()
        assert a.dtype == b.dtype
        np.testing.assert_array_equal(a.view(np.uint8), b.view(np.uint8), err_msg=name)   # bit for bit, NaN included
    a, b = got_te["te"].cpu().numpy(), exp_te.cpu().numpy()
    np.testing.assert_array_equal(a.view(np.uint8), b.view(np.uint8))
    assert int(flag.item()) == int(flag2.item()) == 1
    # outside a pass: only the asking consumer's columns, same values
    alone, _ = index.image_lookup(te_cons, [key], [valid], fold=fold)
    np.testing.assert_array_equal(alone["te"].cpu().numpy().view(np.uint8), b.view(np.uint8))


def test_image_lookup_by_group_ids_through_the_c_abi():
    """gid_out of one call feeds gid_in of the next (no second probe): same columns."""
    from nvtabular_amd import _lib
    from nvtabular_amd import kernels as K

    dev = torch.device("cuda")
    rng = np.random.default_rng(9)
    ids, rows = _keys(rng, 50_000, 4_000)
    index = K.FlatIndex(torch.from_numpy(ids).to(dev))
    key = torch.from_numpy(rows).to(dev)
    vals = torch.from_numpy(rng.normal(size=ids.size)).to(dev)
    image = torch.empty(ids.size * 8, dtype=torch.uint8, device=dev)
    K.image_pack(image, 8, [(vals, torch.float32, 4)], ids.size)
    lib = _lib.load()

    def call(keys_ptr, gid_in, gid_out, out):
        K.check(lib.nvt_flat_lookup_image(
            keys_ptr, K.dtype_code(torch.int32), None, key.numel(), index.aux.data_ptr(),
            index.table.data_ptr(), index.capacity, 0, gid_in, gid_out, image.data_ptr(), 8, 1,
            _lib.ptr_array([out.data_ptr()]), _lib.ptr_array([None]), (C.c_uint32 * 1)(4),
            (C.c_uint32 * 1)(4), (C.c_uint64 * 1)(0), None, K.stream_ptr()), "nvt_flat_lookup_image")

    gid = torch.empty(key.numel(), dtype=torch.int32, device=dev)
    o1 = torch.empty(key.numel(), dtype=torch.float32, device=dev)
    o2 = torch.empty_like(o1)
    call(key.data_ptr(), None, gid.data_ptr(), o1)
    call(None, gid.data_ptr(), None, o2)
    exp = vals.to(torch.float32)[torch.from_numpy(np.searchsorted(ids, rows)).to(dev)]
    assert torch.equal(o1, exp) and torch.equal(o2, exp)
    assert torch.equal(gid.cpu(), torch.from_numpy(np.searchsorted(ids, rows).astype(np.int32)))


@pytest.mark.parametrize("order", ["te_jg", "jg_te"])
@pytest.mark.parametrize("nparts", [1, 3])
def test_workflow_serves_both_operators_from_one_launch_per_partition(tmp_path, order, nparts):
    """TargetEncoding + JoinGroupby on one key column: one image launch per transformed
    partition (not one per operator), results equal to the oracle and to the engine with
    NVT_LOOKUP_IMAGES off (the per-operator kernels)."""
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K
    from nvtabular_amd import ops

    rng = np.random.default_rng(12)
    n = 120_000
    ids, rows = _keys(rng, n, 8_000)
    df = pd.DataFrame({"k": rows, "x": rng.normal(size=n), "y": (rng.random(n) < 0.3).astype("float32")})
    df.loc[rng.random(n) < 0.05, "x"] = np.nan
    cuts = np.linspace(0, n, nparts + 1).astype(int)
    parts = [df.iloc[a:b].reset_index(drop=True) for a, b in zip(cuts[:-1], cuts[1:])]
    stats = ["count", "sum", "mean", "std"]

    def build(tag):
        te = ["k"] >> ops.TargetEncoding(["y", "x"], out_path=str(tmp_path / f"te{tag}"), kfold=5,
                                         fold_seed=42, p_smooth=20)
        jg = ["k"] >> ops.JoinGroupby(out_path=str(tmp_path / f"jg{tag}"), stats=stats, cont_cols=["x", "y"])
        return nvt.Workflow(te + jg if order == "te_jg" else jg + te)

    wf = build("a").fit(nvt.Dataset(parts))
    before, other = K.STATS.get("image_lookups", 0), K.STATS.get("flat_lookups", 0)
    got = wf.transform(nvt.Dataset(parts)).to_ddf().compute()
    assert K.STATS["image_lookups"] == before + nparts
    assert K.STATS.get("flat_lookups", 0) == other   # and no per-operator probe next to it
    K.LOOKUP_IMAGES = False   # the SAME fitted statistics through the per-operator kernels
    try:
        before = K.STATS.get("image_lookups", 0)
        ref = wf.transform(nvt.Dataset(parts)).to_ddf().compute()
        assert K.STATS.get("image_lookups", 0) == before
    finally:
        K.LOOKUP_IMAGES = True
    assert list(got.columns) == list(ref.columns)
    for c in got.columns:
        assert got[c].dtype == ref[c].dtype
        np.testing.assert_array_equal(got[c].to_numpy().view(np.uint8), ref[c].to_numpy().view(np.uint8), err_msg=c)
    cats = O.join_groupby_fit([p.copy() for p in parts], ["k"], ["x", "y"], stats, str(tmp_path / "c"))
    exp_j = O.join_groupby_transform(df.copy(), ["k"], cats)
    st, means = O.target_encoding_fit([p.copy() for p in parts], ["k"], ["y", "x"], str(tmp_path / "c2"),
                                      kfold=5, fold_seed=42)
    exp_t = pd.concat([O.target_encoding_transform(p[["k", "y", "x"]].copy(), ["k"], ["y", "x"], st, means,
                                                   kfold=5, fold_seed=42, p_smooth=20) for p in parts],
                      ignore_index=True)
    for exp in (exp_j, exp_t):
        for c in exp.columns:
            if c in ("k", "x", "y"):
                continue
            if c.endswith("_count"):
                np.testing.assert_array_equal(got[c].to_numpy(), exp[c].to_numpy())
            else:
                np.testing.assert_allclose(got[c].to_numpy().astype("float64"), exp[c].to_numpy().astype("float64"),
                                           rtol=2e-5, atol=1e-6, err_msg=c)
    # a frame with unseen keys: float columns NaN / mean, the int32 count column raises as the
    # reference's astype(int32) does (join_groupby.py:214)
    other = pd.DataFrame({"k": np.array([int(rows[0]), 2**31 - 2], dtype=np.int32), "x": [0.0, 0.0],
                          "y": np.zeros(2, dtype="float32")})
    with pytest.raises(ValueError, match="unseen categories"):
        wf.transform(nvt.Dataset(other)).to_ddf().compute()


def test_per_operator_kernels_still_serve_a_fit_without_images(tmp_path, monkeypatch):
    """NVT_LOOKUP_IMAGES=0 (the per-operator path: transform records in the fit, nvt_flat_lookup_te /
    nvt_flat_lookup_gather in the transform) stays a complete, oracle-equal path."""
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K
    from nvtabular_amd import ops

    monkeypatch.setattr(K, "LOOKUP_IMAGES", False)
    rng = np.random.default_rng(4)
    n = 90_000
    ids, rows = _keys(rng, n, 6_000)
    df = pd.DataFrame({"k": rows, "x": rng.normal(size=n), "y": (rng.random(n) < 0.3).astype("float32")})
    stats = ["count", "sum", "mean", "std"]
    te = ["k"] >> ops.TargetEncoding("y", out_path=str(tmp_path / "te"), kfold=5, fold_seed=42, p_smooth=20)
    jg = ["k"] >> ops.JoinGroupby(out_path=str(tmp_path / "jg"), stats=stats, cont_cols=["x"])
    wf = nvt.Workflow(te + jg).fit(nvt.Dataset(df))
    before, flat = K.STATS.get("image_lookups", 0), K.STATS.get("flat_lookups", 0)
    got = wf.transform(nvt.Dataset(df)).to_ddf().compute()
    assert K.STATS.get("image_lookups", 0) == before and K.STATS.get("flat_lookups", 0) == flat + 2
    cats = O.join_groupby_fit([df.copy()], ["k"], ["x"], stats, str(tmp_path / "c"))
    exp_j = O.join_groupby_transform(df.copy(), ["k"], cats)
    st, means = O.target_encoding_fit([df.copy()], ["k"], ["y"], str(tmp_path / "c2"), kfold=5, fold_seed=42)
    exp_t = O.target_encoding_transform(df[["k", "y"]].copy(), ["k"], ["y"], st, means, kfold=5, fold_seed=42,
                                        p_smooth=20)
    np.testing.assert_array_equal(got["k_count"].to_numpy(), exp_j["k_count"].to_numpy())
    for c in ("k_x_sum", "k_x_mean", "k_x_std"):
        np.testing.assert_allclose(got[c].to_numpy().astype("float64"), exp_j[c].to_numpy().astype("float64"),
                                   rtol=2e-5, atol=1e-6, err_msg=c)
    np.testing.assert_allclose(got["TE_k_y"].to_numpy(), exp_t["TE_k_y"].to_numpy(), rtol=1e-5, atol=1e-6)


def test_more_than_24_outputs_on_one_key_column(tmp_path):
    """a JoinGroupby with many columns next to a TargetEncoding with many targets
    on ONE key column needs more than the 24 output columns nvt_flat_lookup_image takes per launch --
    the lookup goes out in several launches instead of raising; results equal the per-operator
    kernels bit for bit and the oracle."""
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K
    from nvtabular_amd import ops

    rng = np.random.default_rng(21)
    n = 100_000
    ids, rows = _keys(rng, n, 7_000)
    conts = [f"x{i}" for i in range(8)]
    df = pd.DataFrame({"k": rows, **{c: rng.normal(size=n) for c in conts},
                       **{f"y{i}": (rng.random(n) < 0.2 + 0.1 * i).astype("float32") for i in range(4)}})
    stats = ["count", "sum", "mean", "std"]
    targets = [f"y{i}" for i in range(4)]
    te = ["k"] >> ops.TargetEncoding(targets, out_path=str(tmp_path / "te"), kfold=5, fold_seed=42, p_smooth=20)
    jg = ["k"] >> ops.JoinGroupby(out_path=str(tmp_path / "jg"), stats=stats, cont_cols=conts)
    wf = nvt.Workflow(te + jg).fit(nvt.Dataset(df))
    got = wf.transform(nvt.Dataset(df)).to_ddf().compute()
    assert len([c for c in got.columns if c.startswith(("TE_", "k_"))]) > K.IMAGE_LOOKUP_MAX_OUTPUTS
    K.LOOKUP_IMAGES = False
    try:
        ref = wf.transform(nvt.Dataset(df)).to_ddf().compute()
    finally:
        K.LOOKUP_IMAGES = True
    assert list(got.columns) == list(ref.columns)
    for c in got.columns:
        np.testing.assert_array_equal(got[c].to_numpy().view(np.uint8), ref[c].to_numpy().view(np.uint8), err_msg=c)
    cats = O.join_groupby_fit([df.copy()], ["k"], conts, stats, str(tmp_path / "c"))
    exp_j = O.join_groupby_transform(df.copy(), ["k"], cats)
    for c in exp_j.columns:
        if c in df.columns:
            continue
        if c.endswith("_count"):
            np.testing.assert_array_equal(got[c].to_numpy(), exp_j[c].to_numpy())
        else:
            np.testing.assert_allclose(got[c].to_numpy().astype("float64"), exp_j[c].to_numpy().astype("float64"),
                                       rtol=2e-5, atol=1e-6, err_msg=c)
