"""PLAIN / uncompressed parquet files written straight from column buffers.

The output half of the parquet path (reference contract: ``Dataset.to_parquet``,
merlin-io via tests/unit/workflow/test_cpu_workflow.py:67-81 and
bench/datasets/tools/nvt_etl.py:154-171).  pyarrow's writer spent the time of a 45 M-row
Criteo partition set in dictionary building, statistics and compression (5 M rows/s); the
columns this engine produces are fixed-width numbers (int64 labels, float64 / float32
normalised values, int32 passthroughs) for which a PLAIN data page is the column buffer itself:

    page = thrift PageHeader | definition levels | values (non-null ones, little endian)

* no nulls: the definition levels are ONE RLE run ("n times 1": a varint and a byte);
* nulls: the hybrid encoding's bit-packed run at bit width 1 IS the Arrow validity bitmap
  (LSB first), so the bitmap bytes are written verbatim behind a run header, and the values are
  compacted on the device before they are copied out (the caller hands over non-null values).

The file layout (magic, row groups of column chunks of pages, thrift-compact FileMetaData
footer) is written by hand: parquet-format's PageHeader / FileMetaData structures in the
thrift compact protocol.  Readers: pyarrow / pandas / the reference's merlin-io read these
files like any other (tests/test_parquet_plain.py reads them back with pyarrow).
Anything else (strings, lists, booleans, requested dtype casts) stays with pyarrow's writer.
"""
from __future__ import annotations

import os
import struct
from typing import List, Optional, Sequence, Tuple

import numpy as np

# parquet physical types / thrift compact type ids
_PQ_TYPE = {np.dtype("int32"): 1, np.dtype("int64"): 2, np.dtype("float32"): 4, np.dtype("float64"): 5}
_CT_BOOL_TRUE, _CT_I32, _CT_I64, _CT_BINARY, _CT_LIST, _CT_STRUCT = 1, 5, 6, 8, 9, 12
PAGE_VALUES = 1 << 20          # values per data page (8 MiB of int64)
ROW_GROUP_ROWS = 1 << 23       # rows per row group


def supported_dtype(dt) -> bool:
    return np.dtype(dt) in _PQ_TYPE


def _varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _zigzag(v: int) -> bytes:
    return _varint((v << 1) ^ (v >> 63))


class _Struct:
    """Minimal thrift compact-protocol struct writer (fields must be added in ascending id)."""

    def __init__(self):
        self.b = bytearray()
        self.last = 0

    def _head(self, fid: int, ctype: int):
        d = fid - self.last
        if 0 < d <= 15:
            self.b.append((d << 4) | ctype)
        else:
            self.b.append(ctype)
            self.b += _zigzag(fid)
        self.last = fid

    def i32(self, fid, v):
        self._head(fid, _CT_I32)
        self.b += _zigzag(int(v))
        return self

    def i64(self, fid, v):
        self._head(fid, _CT_I64)
        self.b += _zigzag(int(v))
        return self

    def binary(self, fid, s):
        s = s.encode() if isinstance(s, str) else bytes(s)
        self._head(fid, _CT_BINARY)
        self.b += _varint(len(s)) + s
        return self

    def struct(self, fid, body: bytes):
        self._head(fid, _CT_STRUCT)
        self.b += body
        return self

    def list(self, fid, etype: int, items: Sequence[bytes]):
        self._head(fid, _CT_LIST)
        n = len(items)
        self.b += bytes([(n << 4) | etype]) if n < 15 else bytes([0xF0 | etype]) + _varint(n)
        for it in items:
            self.b += it
        return self

    def done(self) -> bytes:
        return bytes(self.b) + b"\x00"


def _page_header(num_values: int, page_bytes: int) -> bytes:
    dph = _Struct().i32(1, num_values).i32(2, 0).i32(3, 3).i32(4, 3).done()  # PLAIN, RLE, RLE
    return _Struct().i32(1, 0).i32(2, page_bytes).i32(3, page_bytes).struct(5, dph).done()


def _def_levels(n: int, valid_bytes: Optional[memoryview]) -> Tuple[bytes, Optional[memoryview]]:
    """Definition levels of one page (max level 1): (prefix bytes, bitmap bytes or None)."""
    if valid_bytes is None:
        body = _varint(n << 1) + b"\x01"          # RLE run: n times the value 1
        return struct.pack("<I", len(body)) + body, None
    groups = (n + 7) // 8
    head = _varint((groups << 1) | 1)             # bit-packed run of `groups` groups of 8 levels
    return struct.pack("<I", len(head) + groups) + head, valid_bytes[:groups]


class PlainParquetWriter:
    """One parquet file of flat int32 / int64 / float32 / float64 columns, every column OPTIONAL
    (like pyarrow writes nullable Arrow columns), PLAIN encoding, no compression.

    With a thread ``pool`` a row group is written by a pool task: a PLAIN page's size is known
    before it is written, so the row group is laid out first and its pages go to their offsets
    with ``os.pwrite`` (which releases the GIL) while the caller stages the next row group or
    another file."""

    def __init__(self, path: str, names: Sequence[str], dtypes: Sequence, pool=None):
        self.path = path
        self.names = list(names)
        self.dtypes = [np.dtype(d) for d in dtypes]
        for d in self.dtypes:
            if d not in _PQ_TYPE:
                raise TypeError(f"PlainParquetWriter: unsupported dtype {d}")
        self.fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        os.pwrite(self.fd, b"PAR1", 0)
        self.pos = 4
        self.pool = pool
        self.row_groups: List[bytes] = []
        self.num_rows = 0
        self.pending = []

    def _plan_column(self, values, valid, n, dt, start):
        """[(offset, bytes-like)] of one column chunk laid out from `start`, and its size."""
        values = np.asarray(values)
        if values.dtype != dt or not values.flags.c_contiguous:
            # never cast here: the buffer may still be the target of an asynchronous device-to-host
            # copy (`ready`), and a silent cast would hide a schema change between row groups
            raise TypeError(f"PlainParquetWriter: column buffer is {values.dtype} "
                            f"(contiguous={values.flags.c_contiguous}), the file's column is {dt}")
        vbytes = memoryview(values).cast("B") if values.size else memoryview(b"")
        vb = memoryview(np.ascontiguousarray(valid)).cast("B") if valid is not None else None
        segs, at = [], start
        done_rows, done_vals = 0, 0
        while True:
            rows = min(PAGE_VALUES, n - done_rows)
            if vb is not None:
                page_valid = vb[done_rows // 8: (done_rows + rows + 7) // 8]
                nv = int(np.unpackbits(np.frombuffer(page_valid, dtype=np.uint8),
                                       bitorder="little")[:rows].sum()) if rows else 0
            else:
                page_valid, nv = None, rows
            prefix, bitmap = _def_levels(rows, page_valid)
            payload = vbytes[done_vals * dt.itemsize: (done_vals + nv) * dt.itemsize]
            body = len(prefix) + (len(bitmap) if bitmap is not None else 0) + len(payload)
            head = _page_header(rows, body) + prefix
            segs.append((at, head))
            at += len(head)
            if bitmap is not None and len(bitmap):
                segs.append((at, bitmap))
                at += len(bitmap)
            if len(payload):
                segs.append((at, payload))
                at += len(payload)
            done_rows += rows
            done_vals += nv
            if done_rows >= n:
                break
        if done_vals != values.size:
            raise ValueError("PlainParquetWriter: the values do not match the validity bitmap "
                             f"({values.size} values, {done_vals} valid rows)")
        return segs, at - start

    def _run(self, segs, ready=None):
        if ready is not None:
            ready()   # (e.g. the event behind the device-to-host copy of these buffers)
        for off, buf in segs:
            mv = memoryview(buf)
            while len(mv):          # (pwrite may write less than asked for)
                k = os.pwrite(self.fd, mv, off)
                mv, off = mv[k:], off + k

    def write_row_group(self, columns, n: int, wait: bool = True, ready=None, stats=None):
        """columns[j] = (values, valid): `values` a 1-D numpy array of the column's dtype holding
        the NON-NULL values in row order, `valid` None or the Arrow validity bitmap (uint8, LSB
        first, >= ceil(n / 8) bytes) of the n rows.  wait=False (with a pool): returns the
        futures of the column writes instead of waiting for them -- the layout is fixed, so later
        row groups (of this or of other files) can be written meanwhile; the caller keeps the
        buffers alive until the futures are done, close() waits for whatever is left.
        ready: called by every column task before it touches its buffers (the VALUES may still
        be in flight when this returns; validity bitmaps must be complete: pages are laid out
        from their popcounts).
        stats[j]: None, or a 2-element numpy array of the column's dtype that holds {min, max} of
        the non-null values by the time close() runs (it may still be in flight now): written as
        the chunk's min / max statistics (NaN / empty chunks: omitted).  The null count of every
        chunk is always written."""
        chunks, plans = [], []
        total = 0
        if len(columns) != len(self.names):
            raise ValueError(f"PlainParquetWriter: {len(columns)} columns for a file of {len(self.names)}")
        for j, ((values, valid), name, dt) in enumerate(zip(columns, self.names, self.dtypes)):
            start = self.pos
            segs, size = self._plan_column(values, valid, n, dt, start)
            plans.append(segs)
            self.pos += size
            nvalid = int(np.asarray(values).size)
            chunks.append(dict(name=name, dt=dt, n=n, size=size, start=start, nulls=n - nvalid,
                               minmax=stats[j] if (stats is not None and nvalid > 0) else None))
            total += size
        futures = []
        if self.pool is not None:
            # ONE task per row group: buffered writes to one file serialise on its inode lock
            # (11 GB/s on the GPU box whatever the thread count, tools/write_probe.py), so
            # threads are spent on DIFFERENT files (6 files: 58 GB/s), not on one file's columns
            futures = [self.pool.submit(self._run, [sg for segs in plans for sg in segs], ready)]
            if wait:
                for f in futures:
                    f.result()
                futures = []
            else:
                self.pending += futures
        else:
            for segs in plans:
                self._run(segs, ready)
        self.row_groups.append((chunks, total, n))   # (thrift structs are built by close(): statistics)
        self.num_rows += n
        return futures

    @staticmethod
    def _chunk_struct(c) -> bytes:
        dt, name = c["dt"], c["name"]
        st = _Struct()
        mm = c["minmax"]
        lo = hi = None
        if mm is not None:
            mm = np.asarray(mm)
            if mm.dtype == dt and mm.size == 2 and not (mm.dtype.kind == "f" and not np.isfinite(mm).all()):
                lo, hi = mm[0:1].tobytes(), mm[1:2].tobytes()
        if hi is not None:
            st.binary(1, hi).binary(2, lo)               # (deprecated pair: signed order, same bytes)
        st.i64(3, c["nulls"])
        if hi is not None:
            st.binary(5, hi).binary(6, lo)               # max_value / min_value
        meta = (_Struct().i32(1, _PQ_TYPE[dt]).list(2, _CT_I32, [_zigzag(0), _zigzag(3)])
                .list(3, _CT_BINARY, [_varint(len(name.encode())) + name.encode()])
                .i32(4, 0).i64(5, c["n"]).i64(6, c["size"]).i64(7, c["size"]).i64(9, c["start"])
                .struct(12, st.done()).done())
        return _Struct().i64(2, c["start"]).struct(3, meta).done()

    def close(self):
        for f in self.pending:
            f.result()
        self.pending = []
        schema = [_Struct().binary(4, "schema").i32(5, len(self.names)).done()]
        for name, dt in zip(self.names, self.dtypes):
            schema.append(_Struct().i32(1, _PQ_TYPE[dt]).i32(3, 1).binary(4, name).done())
        groups = [_Struct().list(1, _CT_STRUCT, [self._chunk_struct(c) for c in chunks]).i64(2, total).i64(3, n).done()
                  for chunks, total, n in self.row_groups]
        # column_orders: TYPE_ORDER for every column (min_value / max_value are only defined with it)
        type_order = _Struct().struct(1, _Struct().done()).done()
        footer = (_Struct().i32(1, 1).list(2, _CT_STRUCT, schema).i64(3, self.num_rows)
                  .list(4, _CT_STRUCT, groups)
                  .binary(6, "nvtabular_amd plain writer")
                  .list(7, _CT_STRUCT, [type_order] * len(self.names)).done())
        self._run([(self.pos, footer + struct.pack("<I", len(footer)) + b"PAR1")])
        os.close(self.fd)
        self.fd = -1

    def abort(self):
        """A failed write: wait for what is in flight, close the descriptor and remove the
        footer-less file (a truncated part file must not be left behind as if it were output)."""
        for f in self.pending:
            try:
                f.result()
            except Exception:
                pass
        self.pending = []
        if self.fd >= 0:
            os.close(self.fd)
            self.fd = -1
        try:
            os.unlink(self.path)
        except OSError:
            pass


# ================================================================================================
# reading side: footer (thrift compact FileMetaData) + column chunks of PLAIN / uncompressed pages
# ================================================================================================
class _TReader:
    """Minimal thrift compact-protocol reader: a struct comes back as {field id: value}, nested
    structs as dicts, lists as Python lists, binaries as bytes."""

    def __init__(self, buf, pos=0):
        self.b, self.p = buf, pos

    def varint(self) -> int:
        v = sh = 0
        while True:
            c = self.b[self.p]
            self.p += 1
            v |= (c & 0x7F) << sh
            if not c & 0x80:
                return v
            sh += 7

    def zigzag(self) -> int:
        v = self.varint()
        return (v >> 1) ^ -(v & 1)

    def value(self, t):
        if t == 1:
            return True
        if t == 2:
            return False
        if t == 3:
            self.p += 1
            return self.b[self.p - 1]
        if t in (4, 5, 6):
            return self.zigzag()
        if t == 7:
            self.p += 8
            return struct.unpack("<d", bytes(self.b[self.p - 8:self.p]))[0]
        if t == 8:
            n = self.varint()
            self.p += n
            return bytes(self.b[self.p - n:self.p])
        if t in (9, 10):
            h = self.b[self.p]
            self.p += 1
            n = h >> 4
            if n == 15:
                n = self.varint()
            et = h & 0x0F
            if et in (1, 2):   # list<bool>: one byte per element
                out = [self.b[self.p + i] == 1 for i in range(n)]
                self.p += n
                return out
            return [self.value(et) for _ in range(n)]
        if t == 11:
            n = self.varint()
            if n == 0:
                return {}
            kv = self.b[self.p]
            self.p += 1
            return {self.value(kv >> 4): self.value(kv & 0x0F) for _ in range(n)}
        if t == 12:
            return self.struct()
        raise ValueError(f"thrift compact: unknown type {t}")

    def struct(self) -> dict:
        out, fid = {}, 0
        while True:
            h = self.b[self.p]
            self.p += 1
            if h == 0:
                return out
            d, t = h >> 4, h & 0x0F
            fid = fid + d if d else self.zigzag()
            out[fid] = self.value(t)


_PQ_NP = {1: np.dtype("int32"), 2: np.dtype("int64"), 4: np.dtype("float32"), 5: np.dtype("float64")}


class PlainParquetFile:
    """Footer of one parquet file and, per row group, the column chunks the hand-written reader
    can take: flat columns (no nesting), physical type INT32 / INT64 / FLOAT / DOUBLE without a
    converted / logical type that changes the meaning of the bits (dates, decimals, unsigned),
    codec UNCOMPRESSED or SNAPPY, values PLAIN or dictionary-encoded (what pandas / pyarrow / cuDF
    write by default), encodings within {PLAIN, PLAIN_DICTIONARY, RLE, BIT_PACKED, RLE_DICTIONARY}.
    ``eligible`` says whether EVERY column of every row group qualifies; otherwise the caller
    reads the file with pyarrow."""

    def __init__(self, path: str):
        self.path = path
        size = os.path.getsize(path)
        with open(path, "rb") as f:
            if size < 12:
                raise ValueError(f"{path}: not a parquet file")
            f.seek(size - 8)
            tail = f.read(8)
            if tail[4:] != b"PAR1":
                raise ValueError(f"{path}: no parquet magic")
            flen = struct.unpack("<I", tail[:4])[0]
            f.seek(size - 8 - flen)
            meta = _TReader(f.read(flen)).struct()
        schema = meta.get(2, [])
        self.num_rows = int(meta.get(3, 0))
        self.eligible, self.why = True, ""
        root_children = int(schema[0].get(5, 0)) if schema else 0
        leaves = schema[1:]
        if len(leaves) != root_children or any(e.get(5) for e in leaves):
            self.eligible, self.why = False, "nested schema"
        self.names = [e.get(4, b"").decode() for e in leaves]
        self.dtypes, self.max_def = [], []
        for e in leaves:
            ptype, rep = e.get(1), e.get(3, 0)
            conv, logical = e.get(6), e.get(10)
            dt = _PQ_NP.get(ptype)
            if dt is None or rep == 2:
                self.eligible, self.why = False, f"column {e.get(4)!r}: physical type {ptype} / repetition {rep}"
            # converted types that reinterpret the integer: DATE 6, TIME 7-8, TIMESTAMP 9-10,
            # UINT 11-14, DECIMAL 5; INT_8 / INT_16 (15 / 16) on a physical INT32 come back as
            # int8 / int16 from pyarrow (legacy writers without a LogicalType): only INT_32 (17)
            # on INT32 and INT_64 (18) on INT64 keep dtype AND bits -- the same rule as the
            # LogicalType INTEGER below; everything else is left to the pyarrow reader
            if conv is not None and conv != {1: 17, 2: 18}.get(ptype):
                self.eligible, self.why = False, f"column {e.get(4)!r}: converted type {conv}"
            if logical is not None:
                # LogicalType union: 10 = INTEGER {1: bitWidth, 2: isSigned}
                integer = logical.get(10) if isinstance(logical, dict) else None
                if not (integer is not None and integer.get(2, True) and
                        integer.get(1, 0) == (32 if ptype == 1 else 64) and len(logical) == 1):
                    self.eligible, self.why = False, f"column {e.get(4)!r}: logical type {logical}"
            self.dtypes.append(dt)
            self.max_def.append(0 if rep == 0 else 1)
        self.row_groups = []
        for rg in meta.get(4, []):
            cols = []
            for cc in rg.get(1, []):
                md = cc.get(3) or {}
                enc = set(md.get(2, []))
                # codec 0 UNCOMPRESSED / 1 SNAPPY; encodings PLAIN 0, PLAIN_DICTIONARY 2, RLE 3,
                # BIT_PACKED 4, RLE_DICTIONARY 8 (nvt_pq_decode_chunk_codec); the chunk starts at its
                # dictionary page when it has one
                ok = (md.get(4) in (0, 1) and enc <= {0, 2, 3, 4, 8} and cc.get(1) in (None, b""))
                if not ok:
                    self.eligible, self.why = False, (f"chunk of {md.get(3)}: codec {md.get(4)}, encodings "
                                                      f"{sorted(enc)}")
                first = int(md.get(9, 0))
                dpo = md.get(11)
                if dpo is not None and 0 < int(dpo) < first:
                    first = int(dpo)
                cols.append(dict(offset=first, size=int(md.get(7, 0)), num_values=int(md.get(5, 0)),
                                 codec=int(md.get(4, 0)), raw_size=int(md.get(6, 0)),
                                 dictionary=bool(dpo is not None or (enc & {2, 8})),
                                 path=[x.decode() for x in md.get(3, [])]))
                if not (0 <= cols[-1]["offset"] and 0 <= cols[-1]["size"] and
                        cols[-1]["offset"] + cols[-1]["size"] <= size):
                    self.eligible, self.why = False, f"chunk of {md.get(3)} lies outside the file"
            if [c["path"] for c in cols] != [[n] for n in self.names]:
                self.eligible, self.why = False, "column chunks do not follow the schema order"
            self.row_groups.append(dict(num_rows=int(rg.get(3, 0)), columns=cols))

    @property
    def num_row_groups(self):
        return len(self.row_groups)


class StagedColumn:
    """One column of a partition in pinned host memory: packed (non-null) values + validity
    bitmap, as nvt_pq_decode_chunk leaves them."""

    __slots__ = ("values", "valid", "rows", "nvalid", "dtype")

    def __init__(self, values, valid, rows, nvalid, dtype):
        self.values, self.valid, self.rows, self.nvalid, self.dtype = values, valid, rows, nvalid, dtype


_TLS = None


def _scratch(nbytes: int) -> bytearray:
    """This thread's read buffer, grown in powers of two."""
    global _TLS
    if _TLS is None:
        import threading

        _TLS = threading.local()
    buf = getattr(_TLS, "buf", None)
    if buf is None or len(buf) < nbytes:
        cap = 1 << 20
        while cap < nbytes:
            cap <<= 1
        buf = _TLS.buf = bytearray(cap)
    return buf


def _scratch2(nbytes: int) -> bytearray:
    """This thread's decompression scratch (dictionary + one page), grown in powers of two."""
    _scratch(1)   # (creates _TLS)
    buf = getattr(_TLS, "buf2", None)
    if buf is None or len(buf) < nbytes:
        cap = 1 << 20
        while cap < nbytes:
            cap <<= 1
        buf = _TLS.buf2 = bytearray(cap)
    return buf


# column chunks decoded by the hand-written reader / left to pyarrow since the process started
# (bench.py's end_to_end entry reports them: a fallback must not be silent)
READER_CHUNKS = {"plain": 0, "pyarrow": 0}


def read_row_groups_staged(pf: PlainParquetFile, groups, columns=None, pool=None, pin=True):
    """{column: StagedColumn} for the concatenation of `groups` (row-group indices) of an eligible
    file.  One task per column: pread of a chunk into a scratch buffer, then nvt_pq_decode_chunk
    (ctypes: GIL released) moves its values and validity bits to their place in the partition's
    (pinned) staging buffers."""
    import ctypes as C

    import torch

    from . import _lib

    lib = _lib.load()
    names = [n for n in pf.names if columns is None or n in columns]
    total = sum(pf.row_groups[g]["num_rows"] for g in groups)
    pinned = bool(pin) and torch.cuda.is_available()   # (host-only processes stage in pageable memory)
    fd = os.open(pf.path, os.O_RDONLY)

    def task(n):
        """One column: its chunks of the row groups one after the other -- packed values behind
        each other, validity bits at the partition's row positions (a thread per COLUMN: two row
        groups may share a bitmap byte)."""
        j = pf.names.index(n)
        dt = pf.dtypes[j]
        vals = torch.empty(total, dtype=getattr(torch, dt.name), pin_memory=pinned)
        valid = None
        if pf.max_def[j]:
            valid = torch.zeros(((total + 63) // 64) * 8 + 8, dtype=torch.uint8, pin_memory=pinned)
        row_at = val_at = 0
        for g in groups:
            cc = pf.row_groups[g]["columns"][j]
            rows = pf.row_groups[g]["num_rows"]
            if rows == 0:
                continue
            buf = _scratch(cc["size"])   # (per thread, reused: a fresh 30 MB bytearray is zero-filled
            got, mv = 0, memoryview(buf)  #  and page-faulted in for every chunk)
            while got < cc["size"]:
                k = os.preadv(fd, [mv[got:cc["size"]]], cc["offset"] + got)
                if k <= 0:
                    raise IOError(f"{pf.path}: short read of column chunk {n}")
                got += k
            cbuf = (C.c_uint8 * len(buf)).from_buffer(buf)
            r, v = C.c_uint64(), C.c_uint64()
            sbuf, sbytes = None, 0
            if cc.get("codec", 0) != 0 or cc.get("dictionary"):
                sbytes = 2 * max(cc.get("raw_size", 0), cc["size"]) + 64
                sraw = _scratch2(sbytes)
                sbuf = (C.c_uint8 * len(sraw)).from_buffer(sraw)
            rc = lib.nvt_pq_decode_chunk_codec(cbuf, cc["size"], cc.get("codec", 0), dt.itemsize, pf.max_def[j],
                                               rows, valid.data_ptr() if valid is not None else None, row_at,
                                               vals.data_ptr() + val_at * dt.itemsize,
                                               (total - val_at) * dt.itemsize, sbuf, sbytes,
                                               C.byref(r), C.byref(v))
            if rc != 0:
                raise _lib.NvtHipError(f"nvt_pq_decode_chunk_codec({pf.path}, {n}, row group {g}): "
                                       f"{lib.nvt_last_error().decode()} (rc {rc})")
            READER_CHUNKS["plain"] += 1
            row_at += rows
            val_at += int(v.value)
        return StagedColumn(vals, valid if (valid is not None and val_at < total) else None, total, val_at, dt)

    futs = []
    try:
        if pool is not None and len(names) > 1:
            futs = [pool.submit(task, n) for n in names]
            cols = [f.result() for f in futs]
        else:
            cols = [task(n) for n in names]
    finally:
        # (after a failed column the others may still be reading: the descriptor is closed -- and
        # its number free for the next open -- only when none of them uses it any more)
        for f in futs:
            f.cancel()
        for f in futs:
            if not f.cancelled():
                f.exception()
        os.close(fd)
    return dict(zip(names, cols))
