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


# ---- reading side: footer + nvt_pq_decode_chunk (host C, no GPU) ------------------------------
def _staged_equals_arrow(staged, table):
    for name in table.column_names:
        s, col = staged[name], table.column(name).combine_chunks()
        valid = np.asarray(col.is_valid())
        assert s.rows == len(col) and s.nvalid == int(valid.sum()), name
        if s.valid is None:
            assert valid.all() or s.nvalid == s.rows, name
        else:
            bits = np.unpackbits(s.valid.numpy(), bitorder="little")[:s.rows].astype(bool)
            np.testing.assert_array_equal(bits, valid, err_msg=name)
        got = s.values.numpy()[:s.nvalid]
        exp = col.drop_null().to_numpy()
        assert got.dtype == exp.dtype, name
        np.testing.assert_array_equal(got.view(np.uint8), exp.view(np.uint8), err_msg=name)   # NaN bits too


def _mixed_table(n, seed=0):
    rng = np.random.default_rng(seed)
    return pa.table({
        "a": pa.array(rng.integers(-5, 1000, n).astype("int32"), mask=rng.random(n) < 0.2),
        "b": pa.array(np.where(rng.random(n) < 0.01, np.nan, rng.normal(size=n))),          # NaN != null
        "c": pa.array(rng.integers(0, 2**40, n), mask=rng.random(n) < 0.5),
        "d": pa.array(rng.normal(size=n).astype("float32"), mask=np.ones(n, bool)),           # all null
        "e": pa.array(rng.integers(0, 9, n).astype("int64")),                                 # no nulls
        "f": pa.array(np.repeat(rng.integers(0, 3, (n + 99) // 100), 100)[:n].astype("int32"),
                      mask=np.repeat(rng.random((n + 999) // 1000) < 0.5, 1000)[:n]),         # long RLE runs
    })


@pytest.mark.parametrize("page_version", ["1.0", "2.0"])
@pytest.mark.parametrize("row_group_size", [100_000, 77_777, 1 << 30])
def test_plain_reader_equals_pyarrow_on_pyarrow_files(tmp_path, page_version, row_group_size):
    """PLAIN / uncompressed files as pyarrow writes them (DataPage v1 and v2, small pages, row
    groups that do not end on byte boundaries of the bitmap): values, nulls and dtypes of the
    hand-written reader equal pyarrow's own reader, row groups concatenated."""
    from concurrent.futures import ThreadPoolExecutor

    t = _mixed_table(300_001)
    path = str(tmp_path / "p.parquet")
    pq.write_table(t, path, use_dictionary=False, compression=None, row_group_size=row_group_size,
                   data_page_version=page_version, data_page_size=32 * 1024)
    pf = PP.PlainParquetFile(path)
    assert pf.eligible, pf.why
    assert pf.names == t.column_names and pf.num_rows == t.num_rows
    groups = list(range(pf.num_row_groups))
    with ThreadPoolExecutor(max_workers=4) as pool:
        staged = PP.read_row_groups_staged(pf, groups, pin=False, pool=pool)
    _staged_equals_arrow(staged, pq.ParquetFile(path).read_row_groups(groups))
    # a column subset, one row group in the middle
    if len(groups) > 2:
        staged = PP.read_row_groups_staged(pf, [1], columns=["c", "a"], pin=False)
        assert set(staged) == {"a", "c"}
        _staged_equals_arrow(staged, pq.ParquetFile(path).read_row_groups([1], columns=["a", "c"]))


@pytest.mark.parametrize("page_version", ["1.0", "2.0"])
@pytest.mark.parametrize("dictionary", [True, False])
@pytest.mark.parametrize("compression", ["snappy", None])
def test_reader_equals_pyarrow_on_default_files(tmp_path, compression, dictionary, page_version):
    """What pandas / pyarrow / cuDF / the reference write by default (round 6: snappy blocks,
    dictionary pages with RLE_DICTIONARY indices -- including the mid-chunk fall-back to PLAIN when a
    dictionary outgrows its page, bit width 0 for a constant column, nulls in either page version):
    every chunk is decoded by nvt_pq_decode_chunk_codec, none is left to pyarrow, and values / nulls /
    dtypes equal pyarrow's own reader bit for bit."""
    t = _mixed_table(250_003, seed=3)
    rng = np.random.default_rng(9)
    t = t.append_column("g", pa.array(np.zeros(t.num_rows, dtype="int32")))                        # width 0
    t = t.append_column("h", pa.array((rng.zipf(1.2, t.num_rows) % 3_000_000).astype("int64"),
                                      mask=rng.random(t.num_rows) < 0.1))                            # big dictionary
    path = str(tmp_path / "d.parquet")
    pq.write_table(t, path, use_dictionary=dictionary, compression=compression, row_group_size=90_001,
                   data_page_version=page_version, data_page_size=24 * 1024)
    pf = PP.PlainParquetFile(path)
    assert pf.eligible, pf.why
    before = dict(PP.READER_CHUNKS)
    groups = list(range(pf.num_row_groups))
    staged = PP.read_row_groups_staged(pf, groups, pin=False)
    _staged_equals_arrow(staged, pq.ParquetFile(path).read_row_groups(groups))
    assert PP.READER_CHUNKS["plain"] == before["plain"] + len(groups) * t.num_columns
    assert PP.READER_CHUNKS["pyarrow"] == before["pyarrow"]
    if dictionary:
        assert any(c["dictionary"] for c in pf.row_groups[0]["columns"])
    assert {c["codec"] for c in pf.row_groups[0]["columns"]} == {1 if compression else 0}


def test_pandas_default_file_goes_through_the_hand_written_reader(tmp_path):
    """`df.to_parquet(path)` with no arguments (snappy + dictionary): eligible."""
    rng = np.random.default_rng(2)
    df = pd.DataFrame({"C1": rng.integers(0, 1000, 50_000).astype("int32"), "I1": rng.normal(size=50_000)})
    path = str(tmp_path / "pandas.parquet")
    df.to_parquet(path)
    pf = PP.PlainParquetFile(path)
    assert pf.eligible, pf.why
    staged = PP.read_row_groups_staged(pf, list(range(pf.num_row_groups)), pin=False)
    _staged_equals_arrow(staged, pq.read_table(path))


@pytest.mark.parametrize("dictionary", [True, False])
def test_codec_decoder_survives_corrupted_chunks(tmp_path, dictionary):
    """A snappy / dictionary chunk with flipped bytes (page headers, snappy tags and offsets, index
    runs, bit widths) is decoded or refused -- never read or written outside its buffers: values,
    bitmap and SCRATCH sit in guarded arrays whose margins must stay untouched."""
    import ctypes as C

    from nvtabular_amd import _lib

    n = 4000
    rng = np.random.default_rng(11 + dictionary)
    arr = pa.array(rng.integers(0, 300 if dictionary else 1 << 40, n).astype("int64"), mask=rng.random(n) < 0.2)
    path = str(tmp_path / "c.parquet")
    pq.write_table(pa.table({"a": arr}), path, use_dictionary=dictionary, compression="snappy",
                   data_page_size=2048)
    pf = PP.PlainParquetFile(path)
    assert pf.eligible
    cc = pf.row_groups[0]["columns"][0]
    raw = bytearray(open(path, "rb").read()[cc["offset"]: cc["offset"] + cc["size"]])
    lib = _lib.load()
    guard = 4096
    sbytes = 2 * max(cc["raw_size"], cc["size"]) + 64
    vals = np.full(n * 8 + 2 * guard, 0xA5, dtype="uint8")
    bm = np.full(n // 8 + 16 + 2 * guard, 0xA5, dtype="uint8")
    scr = np.full(sbytes + 2 * guard, 0xA5, dtype="uint8")
    r, v = C.c_uint64(), C.c_uint64()
    seen = set()
    for it in range(3000):
        buf = bytearray(raw)
        if it:
            for _ in range(int(rng.integers(1, 4))):
                at = int(rng.integers(0, len(buf)))
                buf[at] = int(rng.integers(0, 256)) if rng.random() < 0.7 else (0xFF if rng.random() < 0.5 else 0x80)
        b = (C.c_uint8 * len(buf)).from_buffer(buf)
        rc = lib.nvt_pq_decode_chunk_codec(b, len(buf), 1, 8, 1, n, bm.ctypes.data + guard, 0,
                                           vals.ctypes.data + guard, n * 8, scr.ctypes.data + guard, sbytes,
                                           C.byref(r), C.byref(v))
        seen.add(rc)
        assert rc in (0, _lib.NVT_EINVAL, _lib.NVT_EUNSUPPORTED)
        assert (vals[:guard] == 0xA5).all() and (vals[-guard:] == 0xA5).all()
        assert (bm[:guard] == 0xA5).all() and (bm[guard + n // 8 + 16:] == 0xA5).all()
        assert (scr[:guard] == 0xA5).all() and (scr[-guard:] == 0xA5).all()
    assert 0 in seen and _lib.NVT_EINVAL in seen


def test_plain_reader_reads_the_plain_writers_files(tmp_path):
    rng = np.random.default_rng(5)
    n = PP.PAGE_VALUES + 4099
    specs = [("a", "int64", 0.0), ("b", "float64", 0.3), ("c", "int32", 0.05), ("e", "int64", 1.0)]
    path = str(tmp_path / "w.parquet")
    w = PP.PlainParquetWriter(path, [s[0] for s in specs], [s[1] for s in specs])
    for _ in range(2):
        cols = []
        for name, dt, nf in specs:
            vals, mask, packed, bitmap = _col(rng, n, dt, nf)
            cols.append((packed, bitmap))
        w.write_row_group(cols, n)
    w.close()
    pf = PP.PlainParquetFile(path)
    assert pf.eligible, pf.why
    staged = PP.read_row_groups_staged(pf, [0, 1], pin=False)
    _staged_equals_arrow(staged, pq.read_table(path))


@pytest.mark.parametrize("kind", ["gzip", "zstd", "delta", "byte_stream_split", "string", "list", "date", "uint"])
def test_files_the_plain_reader_leaves_to_pyarrow(tmp_path, kind):
    n = 1000
    rng = np.random.default_rng(1)
    x = rng.integers(0, 5, n).astype("int32")
    kw = dict(use_dictionary=False, compression=None)
    t = pa.table({"x": x})
    if kind in ("gzip", "zstd"):
        if not pa.Codec.is_available(kind):
            pytest.skip(f"pyarrow without {kind}")
        kw["compression"] = kind
    elif kind == "delta":
        kw["column_encoding"] = {"x": "DELTA_BINARY_PACKED"}
    elif kind == "byte_stream_split":
        t = pa.table({"x": x.astype("float32")})
        kw["column_encoding"] = {"x": "BYTE_STREAM_SPLIT"}
    elif kind == "string":
        t = pa.table({"x": x, "s": pa.array([str(v) for v in x])})
    elif kind == "list":
        t = pa.table({"x": x, "l": pa.array([[int(v)] for v in x])})
    elif kind == "date":
        t = pa.table({"x": pa.array(x, type=pa.date32())})
    elif kind == "uint":
        t = pa.table({"x": pa.array(x.astype("uint32"))})
    path = str(tmp_path / "q.parquet")
    pq.write_table(t, path, **kw)
    pf = PP.PlainParquetFile(path)
    assert not pf.eligible and pf.why


def test_decode_chunk_rejects_what_it_does_not_handle(tmp_path):
    """The C decoder's own checks (a footer can lie): truncated chunk, wrong row count."""
    import ctypes as C

    from nvtabular_amd import _lib

    t = pa.table({"a": pa.array(np.arange(5000, dtype="int32"), mask=np.arange(5000) % 7 == 0)})
    path = str(tmp_path / "r.parquet")
    pq.write_table(t, path, use_dictionary=False, compression=None)
    pf = PP.PlainParquetFile(path)
    cc = pf.row_groups[0]["columns"][0]
    raw = open(path, "rb").read()[cc["offset"]: cc["offset"] + cc["size"]]
    lib = _lib.load()
    vals = np.zeros(5000, dtype="int32")
    bm = np.zeros(5000 // 8 + 16, dtype="uint8")
    r, v = C.c_uint64(), C.c_uint64()

    def call(buf, rows):
        b = (C.c_uint8 * len(buf)).from_buffer_copy(buf)
        return lib.nvt_pq_decode_chunk(b, len(buf), 4, 1, rows, bm.ctypes.data, 0, vals.ctypes.data, vals.nbytes,
                                       C.byref(r), C.byref(v))

    assert call(raw, 5000) == 0 and r.value == 5000 and v.value == 5000 - 715
    assert call(raw[: len(raw) // 2], 5000) == _lib.NVT_EINVAL
    assert call(raw, 4000) == _lib.NVT_EINVAL      # more rows in the pages than announced
    assert call(raw, 6000) == _lib.NVT_EINVAL      # fewer


@pytest.mark.parametrize("page_version", ["1.0", "2.0"])
@pytest.mark.parametrize("max_def", [0, 1])
def test_decode_chunk_survives_corrupted_pages(tmp_path, page_version, max_def):
    """A chunk with flipped / overwritten bytes (page headers, level runs, lengths) is decoded or
    refused -- never read or written outside its buffers: the outputs sit in guarded arrays whose
    margins must stay untouched, and negative / huge header fields must not wrap the bounds checks."""
    import ctypes as C

    from nvtabular_amd import _lib

    n = 3000
    rng = np.random.default_rng(5 + max_def)
    arr = pa.array(rng.integers(0, 1 << 30, n).astype("int64"), mask=(rng.random(n) < 0.3) if max_def else None)
    t = pa.table({"a": arr}) if max_def else pa.Table.from_arrays(
        [arr], schema=pa.schema([pa.field("a", pa.int64(), nullable=False)]))
    path = str(tmp_path / "f.parquet")
    pq.write_table(t, path, use_dictionary=False, compression=None, data_page_version=page_version,
                   data_page_size=4096)
    pf = PP.PlainParquetFile(path)
    assert pf.eligible and pf.max_def == [max_def]
    cc = pf.row_groups[0]["columns"][0]
    raw = bytearray(open(path, "rb").read()[cc["offset"]: cc["offset"] + cc["size"]])
    lib = _lib.load()
    guard = 4096
    vals = np.full(n * 8 + 2 * guard, 0xA5, dtype="uint8")
    bm = np.full(n // 8 + 16 + 2 * guard, 0xA5, dtype="uint8")
    r, v = C.c_uint64(), C.c_uint64()
    seen = set()
    for it in range(4000):
        buf = bytearray(raw)
        if it:
            for _ in range(int(rng.integers(1, 4))):
                # most damage goes to the first bytes of a page (its header) and to the level runs
                at = int(rng.integers(0, len(buf))) if rng.random() < 0.5 else int(rng.integers(0, 64)) + \
                    int(rng.integers(0, len(buf) // 4096 + 1)) * 4096 % max(1, len(buf) - 64)
                buf[at] = int(rng.integers(0, 256)) if rng.random() < 0.7 else (0xFF if rng.random() < 0.5 else 0x80)
        b = (C.c_uint8 * len(buf)).from_buffer(buf)
        rc = lib.nvt_pq_decode_chunk(b, len(buf), 8, max_def, n, bm.ctypes.data + guard, 0,
                                     vals.ctypes.data + guard, n * 8, C.byref(r), C.byref(v))
        seen.add(rc)
        assert rc in (0, _lib.NVT_EINVAL, _lib.NVT_EUNSUPPORTED)
        assert (vals[:guard] == 0xA5).all() and (vals[-guard:] == 0xA5).all()
        assert (bm[:guard] == 0xA5).all() and (bm[guard + n // 8 + 16:] == 0xA5).all()
        if rc == 0:
            assert r.value == n and v.value <= n
    assert 0 in seen and _lib.NVT_EINVAL in seen


def test_decode_chunk_header_fields_cannot_wrap_the_bounds_checks():
    """Hand-made page headers with negative counts / lengths: a negative num_values must not pass
    `rows + num_values <= expected` by wrapping (its level run would then fill the bitmap far past its
    end), a negative definition_levels_byte_length must not move the value pointer in front of the
    page."""
    import ctypes as C

    from nvtabular_amd import _lib
    from nvtabular_amd.parquet_plain import _Struct, _varint

    lib = _lib.load()
    one = np.arange(1, dtype="int64").tobytes()

    def v1_page(num_values, levels, values):
        body = (len(levels).to_bytes(4, "little") + levels if levels is not None else b"") + values
        dph = _Struct().i32(1, num_values).i32(2, 0).i32(3, 3).i32(4, 3).done()
        return _Struct().i32(1, 0).i32(2, len(body)).i32(3, len(body)).struct(5, dph).done() + body

    def v2_page(num_values, def_bytes, body):
        dph = (_Struct().i32(1, num_values).i32(2, 0).i32(3, num_values).i32(4, 0).i32(5, def_bytes)
               .i32(6, 0).done())
        return _Struct().i32(1, 3).i32(2, len(body)).i32(3, len(body)).struct(8, dph).done() + body

    def call(chunk, max_def, rows):
        vals = np.zeros(64, dtype="int64")
        bm = np.zeros(64, dtype="uint8")
        r, v = C.c_uint64(), C.c_uint64()
        b = (C.c_uint8 * len(chunk)).from_buffer_copy(chunk)
        return lib.nvt_pq_decode_chunk(b, len(chunk), 8, max_def, rows, bm.ctypes.data, 0, vals.ctypes.data,
                                       vals.nbytes, C.byref(r), C.byref(v))

    rle_one = _varint(1 << 1) + b"\x01"                      # one row, level 1
    assert call(v1_page(1, rle_one, one) + v1_page(1, rle_one, one), 1, 2) == 0
    huge_run = _varint((1 << 40) << 1) + b"\x01"             # 2^40 rows "valid"
    assert call(v1_page(1, rle_one, one) + v1_page(-1, huge_run, one), 1, 2) == _lib.NVT_EINVAL
    assert call(v1_page(1, None, one) + v1_page(-1, None, one), 0, 2) == _lib.NVT_EINVAL
    assert call(v2_page(2, 0, one + one), 0, 2) == 0
    assert call(v2_page(2, -16, one + one), 0, 2) == _lib.NVT_EINVAL
    assert call(v2_page(2, 1 << 20, one + one), 0, 2) == _lib.NVT_EINVAL


def test_plain_reader_required_columns_and_empty_files(tmp_path):
    """REQUIRED columns (max definition level 0: no levels in the pages), a file without rows, a
    file of many tiny row groups."""
    n = 10_007
    rng = np.random.default_rng(2)
    schema = pa.schema([pa.field("r", pa.int64(), nullable=False), pa.field("o", pa.float32(), nullable=True)])
    t = pa.table({"r": rng.integers(0, 10**12, n), "o": pa.array(rng.normal(size=n).astype("float32"),
                                                               mask=rng.random(n) < 0.4)}, schema=schema)
    path = str(tmp_path / "req.parquet")
    pq.write_table(t, path, use_dictionary=False, compression=None, row_group_size=1001, data_page_size=2048)
    pf = PP.PlainParquetFile(path)
    assert pf.eligible, pf.why
    assert pf.max_def == [0, 1] and pf.num_row_groups == 10
    staged = PP.read_row_groups_staged(pf, list(range(pf.num_row_groups)), pin=False)
    assert staged["r"].valid is None
    _staged_equals_arrow(staged, pq.read_table(path))
    # no rows at all
    path0 = str(tmp_path / "empty.parquet")
    pq.write_table(t.slice(0, 0), path0, use_dictionary=False, compression=None)
    pf0 = PP.PlainParquetFile(path0)
    assert pf0.num_rows == 0
    if pf0.eligible and pf0.num_row_groups:
        st0 = PP.read_row_groups_staged(pf0, list(range(pf0.num_row_groups)), pin=False)
        assert all(c.rows == 0 for c in st0.values())


def test_plain_reader_property_random_files(tmp_path):
    """Random row counts, null patterns (incl. long runs), page sizes, page versions and row-group
    sizes: the hand-written reader equals pyarrow's on every file (hypothesis-style sweep with a
    fixed seed: 40 files)."""
    rng = np.random.default_rng(20260924)
    for it in range(40):
        n = int(rng.integers(1, 60_000))
        kinds = rng.integers(0, 4, 3)
        cols = {}
        for j, kind in enumerate(kinds):
            dt = ["int32", "int64", "float32", "float64"][int(rng.integers(0, 4))]
            vals = (rng.normal(size=n) * 1e3).astype(dt)
            if kind == 0:
                mask = None
            elif kind == 1:
                mask = rng.random(n) < rng.random()
            elif kind == 2:   # long runs of nulls / non-nulls (RLE runs in the levels)
                run = int(rng.integers(1, 5000))
                mask = np.repeat(rng.random((n + run - 1) // run) < 0.5, run)[:n]
            else:
                mask = np.ones(n, bool) if rng.random() < 0.5 else np.zeros(n, bool)
            cols[f"c{j}"] = pa.array(vals, mask=mask)
        t = pa.table(cols)
        path = str(tmp_path / f"r{it}.parquet")
        pq.write_table(t, path, use_dictionary=False, compression=None,
                       row_group_size=int(rng.integers(1, n + 1)) if rng.random() < 0.7 else n,
                       data_page_version=["1.0", "2.0"][int(rng.integers(0, 2))],
                       data_page_size=int(rng.integers(64, 1 << 16)))
        pf = PP.PlainParquetFile(path)
        assert pf.eligible, pf.why
        groups = list(range(pf.num_row_groups))
        if len(groups) > 64:     # (keep the sweep fast: a window of row groups)
            lo = int(rng.integers(0, len(groups) - 64))
            groups = groups[lo:lo + 64]
        staged = PP.read_row_groups_staged(pf, groups, pin=False)
        _staged_equals_arrow(staged, pq.ParquetFile(path).read_row_groups(groups))


def test_plain_writer_statistics(tmp_path):
    """Every chunk carries its null count; with {min, max} handed over (to_parquet(statistics=True))
    also min / max, readable by pyarrow (row-group pruning of downstream readers); NaN-only /
    all-null chunks carry none."""
    rng = np.random.default_rng(8)
    n = 5000
    a = rng.integers(-1000, 1000, n).astype("int64")
    b = rng.normal(size=n).astype("float32")
    mask_b = rng.random(n) >= 0.25
    path = str(tmp_path / "s.parquet")
    w = PP.PlainParquetWriter(path, ["a", "b", "c"], ["int64", "float32", "float64"])
    w.write_row_group([(a, None), (b[mask_b], np.packbits(mask_b, bitorder="little")),
                       (np.zeros(0, "float64"), np.zeros((n + 7) // 8, "uint8"))], n,
                      stats=[np.array([a.min(), a.max()], "int64"),
                             np.array([b[mask_b].min(), b[mask_b].max()], "float32"), None])
    w.write_row_group([(a[:10], None), (b[:10], None), (np.full(10, np.nan), None)], 10,
                      stats=[None, None, np.array([np.nan, np.nan], "float64")])
    w.close()
    md = pq.read_metadata(path)
    s = md.row_group(0).column(0).statistics
    assert s.has_min_max and s.min == a.min() and s.max == a.max() and s.null_count == 0
    s = md.row_group(0).column(1).statistics
    assert s.has_min_max and s.min == b[mask_b].min() and s.max == b[mask_b].max()
    assert s.null_count == int((~mask_b).sum())
    s = md.row_group(0).column(2).statistics
    assert s.null_count == n and not s.has_min_max
    for j in range(3):
        s = md.row_group(1).column(j).statistics
        assert s.null_count == 0 and not s.has_min_max
    t = pq.read_table(path, filters=[("a", ">", int(a.max()) + 5)])   # pruned by the statistics
    assert t.num_rows == 0
    assert pq.read_table(path).num_rows == n + 10
    pf = PP.PlainParquetFile(path)
    assert pf.eligible, pf.why
