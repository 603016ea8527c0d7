"""Key directory + one-pass record images (include/nvt_hip.h "key directory": nvt_keydir_build /
nvt_keydir_lookup_image / nvt_image_build): the transform side of JoinGroupby
(join_groupby.py:198-217) and TargetEncoding (target_encoding.py:341-371) on the sort path's groups
with one random line of HBM per row.  Equal to numpy's searchsorted on the key list, bit for bit
to the flat-table lookup and the per-operator image kernels, and end to end to the same workflow
with both switched off."""
import ctypes as C

import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lookup(ids, rows, valid=None, key_offset=0, null_group=-1, load=1.0, dtype=np.int32):
    """group id per row through nvt_keydir_build + nvt_keydir_lookup_image (records = the group id)."""
    from nvtabular_amd import _lib
    from nvtabular_amd import kernels as K
    from nvtabular_amd.device import pack_bitmap_device

    dev = torch.device("cuda")
    lib = _lib.load()
    n = ids.size
    keys32 = torch.from_numpy(ids).to(dev)
    B = max(1, int(n / load))
    d = torch.empty(4 * (B + 1), dtype=torch.int32, device=dev)
    K.check(lib.nvt_keydir_build(keys32.data_ptr(), n, B, d.data_ptr(), K.stream_ptr()), "nvt_keydir_build")
    # directory == the definition: `first` monotone from 0 to n, then the bucket's first keys
    dh = d.cpu().numpy().reshape(B + 1, 4)
    first = dh[:, 0].astype(np.int64) & 0xFFFFFFFF
    assert first[0] == 0 and first[-1] == n and np.all(np.diff(first) >= 0)
    cnt = np.diff(np.append(first, n))
    has = cnt > 0
    np.testing.assert_array_equal(dh[has, 1], ids[first[has]])
    two = cnt > 1
    np.testing.assert_array_equal(dh[two, 2], ids[first[two] + 1])
    np.testing.assert_array_equal(dh[has & ~two, 2], dh[has & ~two, 1])
    recs = n + (1 if null_group >= 0 else 0)
    image = torch.arange(recs, dtype=torch.int64, device=dev)          # record g = the int64 g
    key = torch.from_numpy(rows.astype(dtype)).to(dev)
    vb = pack_bitmap_device(torch.from_numpy(valid).to(dev)) if valid is not None else None
    out = torch.empty(rows.size, dtype=torch.int64, device=dev)
    flag = torch.zeros(1, dtype=torch.int64, device=dev)
    miss = (C.c_uint64 * 1)((-1) & 0xFFFFFFFFFFFFFFFF)
    K.check(lib.nvt_keydir_lookup_image(
        key.data_ptr(), K.dtype_code(key.dtype), K.ptr(vb), rows.size, d.data_ptr(), B, keys32.data_ptr(), n,
        key_offset, null_group, image.data_ptr(), 8, 1, _lib.ptr_array([out.data_ptr()]),
        _lib.ptr_array([None]), (C.c_uint32 * 1)(0), (C.c_uint32 * 1)(8), miss, flag.data_ptr(),
        K.stream_ptr()), "nvt_keydir_lookup_image")
    return out.cpu().numpy(), int(flag.item())


def _expect(ids, rows):
    pos = np.searchsorted(ids, rows)
    pos_c = np.minimum(pos, ids.size - 1)
    return np.where(ids[pos_c] == rows, pos_c, -1)


@pytest.mark.parametrize("load", [0.25, 1.0, 3.0, 64.0])
def test_keydir_lookup_equals_searchsorted(load):
    rng = np.random.default_rng(11)
    ids = np.unique(rng.integers(-(2**31), 2**31 - 1, 70_001, dtype=np.int64).astype(np.int32))
    rows = np.concatenate([ids[rng.integers(0, ids.size, 200_000)],
                           rng.integers(-(2**31), 2**31 - 1, 5_000, dtype=np.int64).astype(np.int32)])
    got, flag = _lookup(ids, rows, load=load)
    np.testing.assert_array_equal(got, _expect(ids, rows))
    assert flag == 1


def test_keydir_edges():
    i32 = np.iinfo(np.int32)
    # one key; the smallest / largest int32 as keys; rows below, between and above the list
    for ids in (np.array([7], dtype=np.int32), np.array([i32.min, -3, 0, 5, i32.max], dtype=np.int32),
                np.array([i32.min, i32.min + 1], dtype=np.int32), np.arange(100, 164, dtype=np.int32)):
        rows = np.array([i32.min, i32.min + 1, -4, -3, 0, 1, 5, 6, 7, 8, 99, 100, 131, 163, 164, i32.max - 1,
                         i32.max], dtype=np.int32)
        for load in (0.3, 1.0, 10.0):
            got, _ = _lookup(ids, rows, load=load)
            np.testing.assert_array_equal(got, _expect(ids, rows), err_msg=f"{ids[:4]} load {load}")
    # dense ids + one far outlier: every key but one shares a bucket (bisection inside the bucket)
    ids = np.concatenate([np.arange(100_000, dtype=np.int32), np.array([2**31 - 5], dtype=np.int32)])
    rng = np.random.default_rng(5)
    rows = np.concatenate([rng.integers(-10, 100_050, 50_000).astype(np.int32),
                           np.array([2**31 - 5, 2**31 - 6, 2**31 - 4], dtype=np.int32)])
    got, _ = _lookup(ids, rows)
    np.testing.assert_array_equal(got, _expect(ids, rows))
    # two clusters far apart
    ids = np.concatenate([np.arange(-2**31 + 10, -2**31 + 5010, dtype=np.int64),
                          np.arange(2**31 - 7000, 2**31 - 1, 2, dtype=np.int64)]).astype(np.int32)
    rows = np.concatenate([ids, ids + 1, np.array([0, 1, -1], dtype=np.int32)]).astype(np.int32)
    got, _ = _lookup(ids, rows)
    np.testing.assert_array_equal(got, _expect(ids, rows))


def test_keydir_int64_keys_nulls_and_the_null_group():
    rng = np.random.default_rng(2)
    base = -7_000_000_000_000
    ids64 = np.unique(base + rng.integers(0, 2**32 - 1, 30_000, dtype=np.int64))
    key_offset = int(ids64[0]) + 2**31                    # list key = column key - key_offset
    ids = (ids64 - key_offset).astype(np.int32)
    rows = np.concatenate([ids64[rng.integers(0, ids64.size, 80_000)],
                           base + rng.integers(-2**33, 2**34, 4_000, dtype=np.int64),
                           np.array([np.iinfo(np.int64).min, np.iinfo(np.int64).max, 0], dtype=np.int64)])
    valid = rng.random(rows.size) > 0.1
    pos = np.searchsorted(ids64, rows)
    pos_c = np.minimum(pos, ids64.size - 1)
    exp = np.where(ids64[pos_c] == rows, pos_c, -1)
    got, _ = _lookup(ids, rows, valid=valid, key_offset=key_offset, dtype=np.int64)
    np.testing.assert_array_equal(got, np.where(valid, exp, -1))
    got, flag = _lookup(ids, rows, valid=valid, key_offset=key_offset, null_group=ids.size, dtype=np.int64)
    np.testing.assert_array_equal(got, np.where(valid, exp, ids.size))
    assert flag == 1


@pytest.mark.parametrize("kfold", [0, 5])
@pytest.mark.parametrize("out_dt", [torch.float32, torch.float64])
def test_one_pass_build_equals_the_per_operator_kernels(kfold, out_dt):
    """nvt_image_build == nvt_jg_image + nvt_te_image into a zeroed image, byte for byte (NaN
    payloads included), with a part that fills fewer records than the image holds."""
    from nvtabular_amd import _lib
    from nvtabular_amd import kernels as K

    dev = torch.device("cuda")
    rng = np.random.default_rng(4)
    g = 70_001
    cnt = torch.from_numpy(rng.integers(1, 50, g).astype(np.int64)).to(dev)
    x = rng.normal(size=(2, g)) * 100
    comp = {"count": cnt, "sum": [torch.from_numpy(x[j].copy()).to(dev) for j in range(2)],
            "sumsq": [torch.from_numpy((x[j] ** 2 + rng.random(g)).copy()).to(dev) for j in range(2)],
            "min": [torch.from_numpy((x[j] - 1).copy()).to(dev) for j in range(2)],
            "max": [torch.from_numpy((x[j] + 1).copy()).to(dev) for j in range(2)]}
    te_size = 8 if out_dt == torch.float64 else 4
    te_off, te_w = 0, (kfold + 1) * te_size
    jg0 = (te_w + 7) & ~7
    outs = [("count", 0, torch.int64, jg0), ("sum", 0, torch.float64, jg0 + 8), ("var", 1, torch.float64, jg0 + 16),
            ("mean", 0, torch.float32, jg0 + 24), ("std", 1, torch.float32, jg0 + 28),
            ("min", 1, torch.float32, jg0 + 32), ("max", 0, torch.float32, jg0 + 36),
            ("count", 0, torch.int32, jg0 + 40)]
    total = jg0 + 44
    stride = K.next_pow2(total) if total <= 64 else (total + 63) & ~63
    kf = max(kfold, 1)
    fcnt = rng.integers(0, 4, (g, kf)).astype(np.int64)
    fsum = rng.normal(size=(g, kf)) * fcnt
    tot_c = torch.from_numpy(fcnt.sum(1)).to(dev)
    tot_s = torch.from_numpy(fsum.sum(1)).to(dev)
    fold_c = torch.from_numpy(fcnt.reshape(-1).copy()).to(dev) if kfold else None
    fold_s = torch.from_numpy(fsum.reshape(-1).copy()).to(dev) if kfold else None
    records, g_te = g + 3, g - 1000
    ref = torch.zeros(records * stride, dtype=torch.uint8, device=dev)
    K.jg_image(ref, stride, comp, outs, g)
    K.te_image(ref, stride, te_off, tot_c, tot_s, fold_c, fold_s, kfold, g_te, 20.0, 0.41, out_dt)
    got = torch.full((records * stride,), 0xAB, dtype=torch.uint8, device=dev)
    parts = [K.jg_image_part(comp, outs, g),
             K.te_image_part(te_off, tot_c, tot_s, fold_c, fold_s, kfold, g_te, 20.0, 0.41, out_dt)]
    arr = (_lib.ImagePart * 2)(*[p[0] for p in parts])
    K.check(_lib.load().nvt_image_build(arr, 2, records, got.data_ptr(), stride, K.stream_ptr()), "nvt_image_build")
    assert torch.equal(got, ref)
    # the target's mean taken from the device ({count, sum}: a fit whose mean is still on its way
    # to the host): the same bits as the host's division
    mom = torch.tensor([1234567.0, 0.41 * 1234567.0 + 0.125, 0.0], dtype=torch.float64, device=dev)
    ym = float(mom[1].item()) / float(mom[0].item())
    ref2 = torch.zeros(records * stride, dtype=torch.uint8, device=dev)
    K.te_image(ref2, stride, te_off, tot_c, tot_s, fold_c, fold_s, kfold, g_te, 20.0, ym, out_dt)
    got2 = torch.empty(records * stride, dtype=torch.uint8, device=dev)
    part = K.te_image_part(te_off, tot_c, tot_s, fold_c, fold_s, kfold, g_te, 20.0, 0.0, out_dt, moments=mom)
    arr = (_lib.ImagePart * 1)(part[0])
    K.check(_lib.load().nvt_image_build(arr, 1, records, got2.data_ptr(), stride, K.stream_ptr()), "nvt_image_build")
    assert torch.equal(got2, ref2)


@pytest.mark.parametrize("nparts", [1, 3])
def test_workflow_outputs_do_not_depend_on_the_lookup_path(tmp_path, nparts):
    """TargetEncoding + JoinGroupby on one key column (null keys included): the key directory +
    one-pass images (default) against the flat table + per-operator image kernels."""
    import nvtabular_amd as nvt
    from nvtabular_amd import kernels as K
    from nvtabular_amd import ops

    rng = np.random.default_rng(8)
    n = 120_000
    k = (rng.random(n) ** 3 * 9_000).astype(np.int64) * 48_271 % (2**31 - 1)
    df = pd.DataFrame({"k": pd.array(k, dtype="Int32"), "y": rng.random(n).astype(np.float32),
                       "x": rng.normal(size=n)})
    df.loc[rng.random(n) < 0.02, "k"] = pd.NA
    cuts = np.linspace(0, n, nparts + 1).astype(int)
    frames = [df.iloc[a:b].reset_index(drop=True) for a, b in zip(cuts[:-1], cuts[1:])]

    te_op = ops.TargetEncoding("y", kfold=5, fold_seed=42, p_smooth=20.0, out_path=str(tmp_path / "te"))
    jg_op = ops.JoinGroupby(cont_cols=["x", "y"], stats=["count", "sum", "mean", "std", "min"],
                            out_path=str(tmp_path / "jg"))
    wf = nvt.Workflow((["k"] >> te_op) + (["k"] >> jg_op))
    wf.fit(nvt.Dataset(frames))
    index = jg_op._device_stats["k"].index
    assert isinstance(index, K.FlatIndex) and index._table is None   # (nothing laid the flat table out)

    def run(keyed, one_pass):
        # the SAME fitted statistics (a second fit may differ in the last bit of a float64 sum: the
        # segmented reduction adds the partial sums of neighbouring waves in arrival order)
        old = K.KEYED_IMAGES, K.ONE_PASS_IMAGES
        K.KEYED_IMAGES, K.ONE_PASS_IMAGES = keyed, one_pass
        index._image = None
        try:
            out = wf.transform(nvt.Dataset(df)).to_ddf().compute()
            assert index._image[3] == keyed
            return out
        finally:
            K.KEYED_IMAGES, K.ONE_PASS_IMAGES = old

    before = K.STATS.get("image_lookups", 0)
    a = run(True, True)
    assert K.STATS.get("image_lookups", 0) > before
    b = run(False, False)
    c = run(True, False)
    assert list(a.columns) == list(b.columns) == list(c.columns)
    for col in a.columns:
        for other in (b, c):
            np.testing.assert_array_equal(a[col].to_numpy().view(np.uint8), other[col].to_numpy().view(np.uint8),
                                          err_msg=col)
