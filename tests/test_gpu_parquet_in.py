"""Parquet in through the hand-written reader (parquet_plain.read_row_groups_staged ->
io.StagedPartition.to_device -> nvt_expand_valid): a Dataset(engine="parquet") partition must be
the same DeviceFrame as through pyarrow, and a workflow fitted on it the same workflow.
Reference contract: merlin.io.Dataset feeding Workflow.fit / transform
(tests/unit/workflow/test_cpu_workflow.py:67-81)."""
import ctypes as C

import numpy as np
import pandas as pd
import pyarrow as pa
import pyarrow.parquet as pq
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4096, 4097, 1_000_003])
@pytest.mark.parametrize("size", [4, 8])
@pytest.mark.parametrize("null_frac", [0.0, 0.3, 1.0])
def test_expand_valid_vs_numpy(n, size, null_frac):
    from nvtabular_amd import kernels as K

    dev = torch.device("cuda")
    rng = np.random.default_rng(n + size)
    valid = rng.random(n) >= null_frac
    dt = np.int32 if size == 4 else np.int64
    packed = rng.integers(1, 2**31 - 1, int(valid.sum())).astype(dt)
    bits = np.packbits(valid, bitorder="little")
    pad = (-len(bits)) % 8
    bm = np.concatenate([bits, np.full(pad, 0xFF, dtype=np.uint8)])   # (garbage behind the column)
    if n % 8:
        bm[n // 8] |= np.uint8((0xFF << (n % 8)) & 0xFF)
    lib = K._lib.load()
    need = C.c_uint64()
    K.check(lib.nvt_expand_valid_ws_bytes(n, C.byref(need)), "ws")
    ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    p_dev = torch.from_numpy(packed if packed.size else np.zeros(1, dtype=dt)).to(dev)
    b_dev = torch.from_numpy(bm).to(dev)
    out = torch.full((n,), -7, dtype=torch.int32 if size == 4 else torch.int64, device=dev)
    K.check(lib.nvt_expand_valid(p_dev.data_ptr(), size, b_dev.data_ptr(), n, out.data_ptr(), ws.data_ptr(),
                                 K.stream_ptr()), "nvt_expand_valid")
    exp = np.zeros(n, dtype=dt)
    exp[valid] = packed
    np.testing.assert_array_equal(out.cpu().numpy(), exp)


def _frame(n, seed=0):
    rng = np.random.default_rng(seed)
    return pa.table({
        "C1": pa.array((rng.zipf(1.3, n) % 5000).astype("int32"), mask=rng.random(n) < 0.1),
        "C2": pa.array(rng.integers(0, 40, n).astype("int64")),
        "I1": pa.array(np.floor(rng.lognormal(2, 2, n)).astype("int32"), mask=rng.random(n) < 0.3),
        "I2": pa.array(np.where(rng.random(n) < 0.02, np.nan, rng.normal(size=n)).astype("float32"),
                       mask=rng.random(n) < 0.05),
        "I3": pa.array(rng.normal(size=n), mask=np.ones(n, bool)),          # all null
    })


@pytest.mark.parametrize("page_version", ["1.0", "2.0"])
@pytest.mark.parametrize("dictionary", [False, True])
@pytest.mark.parametrize("compression", [None, "snappy"])
def test_parquet_dataset_through_the_plain_reader_equals_pyarrow(tmp_path, page_version, dictionary, compression,
                                                                 monkeypatch):
    """{snappy, none} x {dictionary on, off} x {v1, v2 pages} (round 6): every partition comes through
    the hand-written reader, frames and the fitted workflow's output equal the pyarrow path's."""
    import nvtabular_amd as nvt
    from nvtabular_amd import io as nio
    from nvtabular_amd import ops

    t = _frame(250_007, 3)
    path = str(tmp_path / "in")
    import os

    os.makedirs(path)
    for i, (a, b) in enumerate([(0, 100_003), (100_003, 250_007)]):
        pq.write_table(t.slice(a, b - a), os.path.join(path, f"part_{i}.parquet"), use_dictionary=dictionary,
                       compression=compression, row_group_size=60_001, data_page_version=page_version,
                       data_page_size=64 * 1024)

    def parts(plain):
        monkeypatch.setattr(nio, "PLAIN_PARQUET_READ", plain)
        ds = nvt.Dataset(path, engine="parquet", row_groups_per_part=2)
        return ds, [f for f in ds.to_iter()]

    seen = []
    orig = nio.StagedPartition.to_device

    def spy(self, device=None):
        seen.append(self.num_rows)
        return orig(self, device)

    monkeypatch.setattr(nio.StagedPartition, "to_device", spy)
    ds_a, a = parts(True)
    assert sum(seen) == t.num_rows and len(seen) == len(a)       # every partition came through the plain reader
    seen.clear()
    ds_b, b = parts(False)
    assert not seen
    assert len(a) == len(b) and sum(len(f) for f in a) == t.num_rows
    for fa, fb in zip(a, b):
        assert list(fa.columns) == list(fb.columns) and len(fa) == len(fb)
        for name in fa.columns:
            ca, cb = fa[name], fb[name]
            assert ca.data.dtype == cb.data.dtype, name
            ma, mb = ca.valid_mask_host(), cb.valid_mask_host()
            if mb is None:
                assert ma is None or ma.all(), name
                mb = np.ones(len(fa), bool)
            else:
                np.testing.assert_array_equal(ma, mb, err_msg=name)
            va, vb = ca.data.cpu().numpy()[mb], cb.data.cpu().numpy()[mb]
            np.testing.assert_array_equal(va.view(np.uint8), vb.view(np.uint8), err_msg=name)
    # the same workflow either way, and equal to the oracle
    def run(plain, tag):
        monkeypatch.setattr(nio, "PLAIN_PARQUET_READ", plain)
        cat = ["C1", "C2"] >> ops.Categorify(out_path=str(tmp_path / f"c{tag}"))
        cont = ["I1", "I2"] >> ops.FillMissing() >> ops.Normalize()
        wf = nvt.Workflow(cat + cont)
        ds = nvt.Dataset(path, engine="parquet", row_groups_per_part=2)
        return wf.fit_transform(ds).to_ddf().compute()

    ga, gb = run(True, "a"), run(False, "b")
    pd.testing.assert_frame_equal(ga, gb)
    host = t.to_pandas()
    cats = O.categorify_fit([host[["C1", "C2"]].copy()], ["C1", "C2"], str(tmp_path / "oc"), tie_break="stable")
    exp = O.categorify_transform(host[["C1", "C2"]].copy(), ["C1", "C2"], cats)
    for c in ("C1", "C2"):
        np.testing.assert_array_equal(ga[c].to_numpy(), exp[c].to_numpy(), err_msg=c)


def test_end_to_end_files_of_the_plain_writer_round_trip(tmp_path):
    """to_parquet (PLAIN writer) -> Dataset(engine="parquet") (PLAIN reader): the frame again."""
    import nvtabular_amd as nvt

    t = _frame(120_000, 9).drop_columns(["I3"])
    df = t.to_pandas()
    nvt.Dataset(t).to_parquet(str(tmp_path / "o"))
    back = nvt.Dataset(str(tmp_path / "o"), engine="parquet").to_ddf().compute()
    for c in df.columns:
        a, b = back[c].to_numpy(dtype="float64", na_value=np.nan), df[c].to_numpy(dtype="float64", na_value=np.nan)
        np.testing.assert_array_equal(a, b, err_msg=c)
