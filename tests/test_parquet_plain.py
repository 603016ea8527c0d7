"""The hand-written PLAIN parquet writer (nvtabular_amd/parquet_plain.py) read back by pyarrow
and pandas: values, nulls, dtypes, row groups, page boundaries, empty files.  Contract:
Dataset.to_parquet output must be readable by any parquet reader
(reference tests/unit/workflow/test_cpu_workflow.py:67-81 reads it back with dask / pandas)."""
import numpy as np
import pandas as pd
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from nvtabular_amd import parquet_plain as PP


def _col(rng, n, dt, null_frac):
    if np.dtype(dt).kind == "f":
        vals = rng.normal(size=n).astype(dt)
    else:
        info = np.iinfo(dt)
        vals = rng.integers(info.min, info.max, n, dtype=dt)
    if null_frac == 0:
        return vals, None, vals, None
    mask = rng.random(n) >= null_frac          # True = valid
    bitmap = np.packbits(mask, bitorder="little")
    return vals, mask, vals[mask], bitmap


@pytest.mark.parametrize("n", [0, 1, 7, 8, 1000, PP.PAGE_VALUES + 13, 2 * PP.PAGE_VALUES])
def test_plain_file_reads_back(tmp_path, n, monkeypatch):
    rng = np.random.default_rng(n)
    specs = [("a", "int64", 0.0), ("b", "float64", 0.3), ("c", "int32", 0.05), ("d", "float32", 0.0),
             ("e", "int64", 1.0)]
    cols, exp = [], {}
    for name, dt, nf in specs:
        vals, mask, packed, bitmap = _col(rng, n, dt, nf)
        cols.append((packed, bitmap))
        exp[name] = pa.array(vals, mask=None if mask is None else ~mask)
    path = str(tmp_path / "f.parquet")
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=3) as pool:   # (even n: pooled pwrites, odd n: serial)
        w = PP.PlainParquetWriter(path, [s[0] for s in specs], [s[1] for s in specs],
                                  pool=pool if n % 2 == 0 else None)
        w.write_row_group(cols, n)
    w.close()
    t = pq.read_table(path)
    assert t.num_rows == n and t.column_names == [s[0] for s in specs]
    for name, dt, _ in specs:
        assert t.schema.field(name).type == pa.from_numpy_dtype(np.dtype(dt))
        assert t.column(name).combine_chunks().equals(exp[name]), name
    md = pq.read_metadata(path)
    assert md.num_row_groups == 1 and md.row_group(0).num_rows == n
    assert md.row_group(0).column(0).compression == "UNCOMPRESSED"
    df = pd.read_parquet(path)
    assert len(df) == n


def test_several_row_groups_and_metadata_collection(tmp_path):
    rng = np.random.default_rng(3)
    path = str(tmp_path / "g.parquet")
    w = PP.PlainParquetWriter(path, ["x", "y"], ["int64", "float64"])
    parts = []
    for n in (1024, 8, 50_000):
        x = rng.integers(0, 1000, n).astype("int64")
        y = rng.normal(size=n)
        mask = rng.random(n) > 0.5
        w.write_row_group([(x, None), (y[mask], np.packbits(mask, bitorder="little"))], n)
        parts.append(pd.DataFrame({"x": x, "y": np.where(mask, y, np.nan)}))
    w.close()
    got = pd.read_parquet(path)
    exp = pd.concat(parts, ignore_index=True)
    pd.testing.assert_frame_equal(got, exp)
    md = pq.read_metadata(path)
    assert md.num_row_groups == 3 and md.num_rows == len(exp)
    # the _metadata summary pyarrow builds from collected footers accepts these files
    md.set_file_path("g.parquet")
    pq.write_metadata(pq.read_schema(path), str(tmp_path / "_metadata"), metadata_collector=[md])
    assert pq.read_metadata(str(tmp_path / "_metadata")).num_rows == len(exp)
    # row-group statistics are absent (not written), filters still work through a dataset read
    import pyarrow.dataset as ds

    sub = ds.dataset(path).to_table(filter=ds.field("x") < 10).to_pandas()
    assert (sub["x"] < 10).all() and len(sub) == int((exp["x"] < 10).sum())


def test_unsupported_dtype_is_refused(tmp_path):
    with pytest.raises(TypeError):
        PP.PlainParquetWriter(str(tmp_path / "h.parquet"), ["s"], ["uint8"])
