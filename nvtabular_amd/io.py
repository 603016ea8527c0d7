"""Dataset: a partitioned table source (re-creation of the slice of
merlin.io.Dataset the hot path touches: SURVEY section 8(b) "Dataset").

Sources: pandas DataFrame, DeviceFrame, pyarrow Table, parquet path(s) or a list
of already-partitioned frames.  ``to_iter()`` yields HBM-resident DeviceFrames;
parquet row groups are decoded by pyarrow straight into Arrow buffers and copied
to the device without going through pandas.
"""
from __future__ import annotations

import glob
import os
from typing import Iterable, List, Optional

import pandas as pd

from .device import DeviceFrame, as_device_frame
from .schema import Schema


DECODE_AHEAD = int(os.environ.get("NVT_DECODE_AHEAD", "3"))  # parquet partitions decoded concurrently ahead of the consumer


class _Collection:
    """What ``Dataset.to_ddf()`` returns: ``.compute()`` gives one pandas frame."""

    def __init__(self, ds: "Dataset"):
        self._ds = ds

    def compute(self, **_):
        return self._ds.compute()

    def head(self, n=5):
        return self.compute().head(n)

    @property
    def columns(self):
        return self._ds.schema.column_names

    @property
    def npartitions(self):
        return self._ds.npartitions


class Dataset:
    def __init__(self, path_or_source, engine=None, cpu=None, part_size=None,
                 part_mem_fraction=None, npartitions=None, names=None, schema=None,
                 row_groups_per_part=1, **kwargs):
        self.cpu = bool(cpu)  # accepted for API compatibility; compute always runs on the GPU
        self.engine = engine
        self._schema = schema
        self._parts_fn = None
        src = path_or_source
        if callable(src):  # lazy partition generator (transformed datasets)
            self._parts_fn = src
            self._n = npartitions
        elif isinstance(src, (pd.DataFrame, DeviceFrame)) or _is_arrow_table(src):
            self._init_frames(_split(src, npartitions or 1))
        elif isinstance(src, (list, tuple)) and src and not isinstance(src[0], (str, os.PathLike)):
            self._init_frames(list(src))
        else:
            self._init_parquet(src, row_groups_per_part, names)

    # ---- sources ---------------------------------------------------------------
    def _init_frames(self, frames):
        self._frames = frames
        self._n = len(frames)
        if self._schema is None:
            f0 = frames[0]
            self._schema = Schema.from_frame(f0.schema if _is_arrow_table(f0) else f0)
        self._parts_fn = lambda columns=None: iter(self._frames)

    def _init_parquet(self, paths, row_groups_per_part, names):
        import pyarrow.parquet as pq

        if isinstance(paths, (str, os.PathLike)):
            paths = [str(paths)]
        files: List[str] = []
        for p in paths:
            p = str(p)
            if os.path.isdir(p):
                files += sorted(glob.glob(os.path.join(p, "*.parquet")))
            elif any(ch in p for ch in "*?["):
                files += sorted(glob.glob(p))
            else:
                files.append(p)
        if not files:
            raise FileNotFoundError(f"no parquet files under {paths}")
        self._files = files
        pieces = []
        for f in files:
            md = pq.ParquetFile(f)
            ng = md.num_row_groups
            for g0 in range(0, ng, row_groups_per_part):
                pieces.append((f, list(range(g0, min(ng, g0 + row_groups_per_part)))))
        self._pieces = pieces
        self._n = len(pieces)
        if self._schema is None:
            self._schema = Schema.from_frame(pq.ParquetFile(files[0]).schema_arrow)

        # files the hand-written reader takes (flat numeric columns; PLAIN or dictionary-encoded
        # values, uncompressed or snappy: parquet_plain.PlainParquetFile); everything else is
        # decoded by pyarrow and counted in parquet_plain.READER_CHUNKS
        plain_files = {}

        def plain_file(f):
            if f not in plain_files:
                pf = None
                if PLAIN_PARQUET_READ:
                    try:
                        from .parquet_plain import PlainParquetFile

                        pf = PlainParquetFile(f)
                        if not pf.eligible:
                            pf = None
                    except Exception:
                        pf = None
                plain_files[f] = pf
            return plain_files[f]

        def read(piece, columns):
            f, groups = piece
            pf = plain_file(f)
            if pf is not None:
                from .parquet_plain import read_row_groups_staged

                try:
                    return StagedPartition(read_row_groups_staged(pf, groups, columns, pool=_plain_read_pool()))
                except Exception as e:  # (a page kind the footer did not announce: pyarrow reads it)
                    from . import _lib

                    if not isinstance(e, _lib.NvtHipError):
                        raise
                    plain_files[f] = None
            from .parquet_plain import READER_CHUNKS

            table = pq.ParquetFile(f).read_row_groups(groups, columns=columns)
            READER_CHUNKS["pyarrow"] += table.num_columns * len(groups)   # (counted: never silent)
            return table

        def gen(columns=None, only=None):
            # decode a few partitions ahead on host threads (one read spreads over the columns of
            # its row groups; several reads in flight keep more of the host cores busy)
            from collections import deque
            from concurrent.futures import ThreadPoolExecutor

            todo = [p for i, p in enumerate(pieces) if only is None or only(i)]
            if len(todo) <= 1:
                for piece in todo:
                    yield read(piece, columns)
                return

            with ThreadPoolExecutor(max_workers=DECODE_AHEAD) as pool:
                window = deque()
                it = iter(todo)
                for piece in it:
                    window.append(pool.submit(read, piece, columns))
                    if len(window) >= DECODE_AHEAD:
                        break
                while window:
                    table = window.popleft().result()
                    nxt = next(it, None)
                    if nxt is not None:
                        window.append(pool.submit(read, nxt, columns))
                    yield table

        self._parts_fn = gen

    # ---- public surface -----------------------------------------------------------
    @property
    def schema(self) -> Schema:
        if callable(self._schema):
            # a transformed dataset: the fitted output schema (embedding sizes ...) is folded
            # together on first use, not before the first partition's kernels are enqueued
            self._schema = self._schema()
        if self._schema is None:
            first = next(iter(self._parts_fn()))
            self._schema = Schema.from_frame(first)
        return self._schema

    @property
    def npartitions(self):
        if self._n is None:
            self._n = sum(1 for _ in self._parts_fn())
        return self._n

    def _host_parts(self, cols, shard):
        """Partitions of this rank.  Under torch.distributed (shard = (rank, world)):

        * a parquet dataset is a GLOBAL list of files / row groups that every rank opens the
          same way: rank r takes every world-th partition and decodes only those;
        * frames handed over in memory (DataFrame, DeviceFrame, Arrow table, list of them) were
          built by THIS process: they are the rank's own shard already and are all kept;
        * a derived dataset (Workflow.transform) defers to its source."""
        if getattr(self, "_pieces", None) is not None:
            if shard is not None:
                yield from self._parts_fn(cols, only=lambda i: i % shard[1] == shard[0])
            else:
                yield from self._parts_fn(cols)
            return
        if getattr(self, "_forwards_shard", False):
            yield from self._parts_fn(cols, shard=shard)
            return
        it = self._parts_fn(cols) if _accepts_columns(self._parts_fn) else self._parts_fn()
        yield from it

    def to_iter(self, columns: Optional[Iterable[str]] = None, shard=None, prefetch=None):
        """Yield DeviceFrame partitions.  shard=(rank, world) keeps every world-th one.

        Parquet sources are double-buffered: a background thread decodes the NEXT row-group
        range with pyarrow (GIL released), stages it in pinned memory and issues the
        host-to-device copies on a side stream while the consumer's kernels run on the
        current stream; the hand-over is a stream event, not a device synchronise."""
        cols = list(columns) if columns is not None else None
        if prefetch is None:
            prefetch = hasattr(self, "_pieces")
        if not prefetch:
            for part in self._host_parts(cols, shard):
                frame, _ = as_device_frame(part)
                if cols is not None:
                    frame = frame[[c for c in cols if c in frame]]
                yield frame
            return
        yield from _prefetch_frames(self._host_parts(cols, shard), cols)

    def to_ddf(self, columns=None, **_):
        return _Collection(self if columns is None else self._select(columns))

    def _select(self, columns):
        cols = list(columns)
        return Dataset(lambda columns=None: (p[cols] for p in self.to_iter(cols)),
                       schema=self.schema.select_by_name(cols), npartitions=self._n)

    def compute(self) -> pd.DataFrame:
        parts = [p.to_pandas() for p in self.to_iter()]
        if not parts:
            return pd.DataFrame(columns=self.schema.column_names)
        return pd.concat(parts, ignore_index=True)

    def head(self, n=5):
        for p in self.to_iter():
            return p.to_pandas().head(n)
        return pd.DataFrame()

    def to_cpu(self):
        self.cpu = True
        return self

    def to_gpu(self):
        self.cpu = False
        return self

    def to_parquet(self, output_path, shuffle=None, out_files_per_proc=None, dtypes=None,
                   cats=None, conts=None, labels=None, preserve_files=False, suffix=".parquet",
                   num_threads=0, compression=None, statistics=False, **_):
        """Write the (transformed) dataset as parquet (merlin.io.Dataset.to_parquet; contract in
        tests/unit/workflow/test_workflow.py:171-187,363-396,444-500 and
        bench/datasets/tools/nvt_etl.py:154-171 of the reference).

        * ``out_files_per_proc=k``: every partition is cut into k pieces, piece j appended to
          ``part_j.parquet`` -- k files per process (``part_{rank*k + j}`` under torchrun).
          ``None``: one file per input partition.
        * ``shuffle``: ``Shuffle.PER_PARTITION`` permutes the rows of each partition (on the
          device, before the copy out); ``Shuffle.PER_WORKER`` additionally permutes each
          output file as a whole (its pieces are held on the host until the end); ``None`` /
          ``False`` keeps the row order.
        * ``dtypes``: {column: dtype} casts applied on the way out.
        * ``compression``: ``None`` (default) lets fixed-width numeric frames take the hand-written
          PLAIN writer (uncompressed pages, no dictionary, no column statistics: files are larger
          than pyarrow's snappy + dictionary output and carry no min / max for predicate
          pushdown -- the price of writing at tens of GB/s); any codec name (``"snappy"``,
          ``"zstd"``, ``"none"`` ...) selects pyarrow's writer with that codec, statistics and
          dictionary pages, as the reference's writer produces.  ``NVT_PLAIN_PARQUET=0`` makes
          pyarrow the default.
        * ``statistics=True``: the PLAIN writer also records min / max of every column chunk
          (computed on the device next to the copy out); the null count of every chunk is always
          written.  pyarrow's writer always writes statistics.
        * writes ``_metadata`` (parquet summary of all row groups), ``_file_list.txt`` and
          ``_metadata.json`` (file stats + cats / conts / labels) next to the data files.
        """
        import itertools
        import json

        import numpy as np
        import pyarrow as pa
        import pyarrow.parquet as pq

        from . import dist

        shuffle = Shuffle.coerce(shuffle)
        pq_kw = {} if compression is None else {"compression": compression}
        os.makedirs(str(output_path), exist_ok=True)
        output_path = str(output_path)
        rank, world = dist.rank(), dist.world_size()
        k = int(out_files_per_proc) if out_files_per_proc else None
        rng = np.random.default_rng()
        writers, held, names, rows_in = {}, {}, {}, {}
        touched = set()
        collector = []

        def fname(j):
            return f"part_{(rank * k + j) if k else (j * world + rank)}{suffix}"

        # Parquet encoding (dictionary + compression) is host work that pyarrow does with the
        # GIL released: file j is always written by lane j % NLANES, in order, so up to NLANES
        # files are encoded at once while the next partition is transformed and copied out.
        import threading
        from concurrent.futures import ThreadPoolExecutor

        NLANES = 4
        lanes = [ThreadPoolExecutor(max_workers=1) for _ in range(NLANES)]
        inflight = threading.BoundedSemaphore(2 * NLANES)  # bounds the host memory queued up
        pending = []

        def write(j, table):
            try:
                w = writers.get(j)
                if w is None:
                    names[j] = fname(j)
                    w = writers[j] = pq.ParquetWriter(os.path.join(output_path, names[j]),
                                                      table.schema, metadata_collector=collector,
                                                      **pq_kw)
                w.write_table(table)
                rows_in[j] = rows_in.get(j, 0) + table.num_rows
            finally:
                inflight.release()

        def emit(j, table):
            if shuffle == Shuffle.PER_WORKER and k:
                held.setdefault(j, []).append(table)
                return
            inflight.acquire()
            pending.append(lanes[j % NLANES].submit(write, j, table))

        def drain():
            for f in pending:
                f.result()  # re-raises a writer's exception here
            pending.clear()

        shard = (rank, world) if world > 1 else None
        parts_iter = iter(self.to_iter(shard=shard))
        first = next(parts_iter, None)
        plain = None
        if first is not None and PLAIN_PARQUET and compression is None and _plain_eligible(first, dtypes) and \
                not (shuffle == Shuffle.PER_WORKER and k):
            # fixed-width numeric columns: PLAIN pages written straight from pinned column
            # buffers (parquet_plain.py) -- no dictionary pass, no compression, no statistics
            plain = _write_plain(itertools.chain([first], parts_iter), output_path, fname, k, shuffle, dtypes,
                                 statistics=bool(statistics))
            parts_iter, first = iter(()), None
        rest = itertools.chain([first], parts_iter) if first is not None else iter(())
        for i, part in enumerate(rest):
            n = len(part)
            if shuffle is not None and n > 1:
                part = part.take_rows(_device_permutation(n, part))
            table = part.to_arrow()  # pinned async copies, no pandas round trip
            if dtypes:
                for c, t in dtypes.items():
                    if c in table.column_names:
                        ci = table.column_names.index(c)  # NOT `i`: that is the partition index
                        table = table.set_column(ci, c, table.column(c).cast(pa.from_numpy_dtype(np.dtype(t))))
            if k is None:
                emit(i, table)
                continue
            bounds = [(n * j) // k for j in range(k + 1)]
            for j in range(k):
                if bounds[j + 1] > bounds[j] or j not in touched:
                    touched.add(j)
                    emit(j, table.slice(bounds[j], bounds[j + 1] - bounds[j]))
        drain()
        for ex in lanes:
            ex.shutdown(wait=True)
        for j, pieces in sorted(held.items()):
            table = pa.concat_tables(pieces)
            if table.num_rows > 1:
                table = table.take(pa.array(rng.permutation(table.num_rows)))
            names[j] = fname(j)
            w = writers[j] = pq.ParquetWriter(os.path.join(output_path, names[j]), table.schema,
                                              metadata_collector=collector, **pq_kw)
            w.write_table(table)
            rows_in[j] = table.num_rows
        schema = None
        order = sorted(writers)
        for j in order:
            schema = schema or writers[j].schema
            writers[j].close()
        # every ParquetWriter appended its FileMetaData on close, in closing order
        for md, j in zip(collector, order):
            md.set_file_path(names[j])
        if plain is not None:
            names, rows_in, order = plain
            collector = []
            for j in order:
                md = pq.read_metadata(os.path.join(output_path, names[j]))
                md.set_file_path(names[j])
                collector.append(md)
            schema = pq.read_schema(os.path.join(output_path, names[order[0]])) if order else None
        if world > 1:
            gathered = [None] * world
            import torch.distributed as td

            td.all_gather_object(gathered, [(names[j], rows_in.get(j, 0)) for j in order])
            td.barrier()
            files = [x for g in gathered for x in g]
        else:
            files = [(names[j], rows_in.get(j, 0)) for j in order]
        if rank == 0 and schema is not None:
            if world > 1:  # summary over every rank's files
                collector = []
                for name, _ in files:
                    md = pq.read_metadata(os.path.join(output_path, name))
                    md.set_file_path(name)
                    collector.append(md)
            pq.write_metadata(schema, os.path.join(output_path, "_metadata"),
                              metadata_collector=collector)
            with open(os.path.join(output_path, "_file_list.txt"), "w") as f:
                f.write(str(len(files)) + "\n")
                for name, _ in files:
                    f.write(name + "\n")
            cols = schema.names
            pick = lambda lst: [{"col_name": c, "index": cols.index(c)} for c in (lst or []) if c in cols]
            meta = {"file_stats": [{"file_name": name, "num_rows": int(nr)} for name, nr in files],
                    "cats": pick(cats), "conts": pick(conts), "labels": pick(labels)}
            with open(os.path.join(output_path, "_metadata.json"), "w") as f:
                json.dump(meta, f)
        return None


PLAIN_PARQUET = os.environ.get("NVT_PLAIN_PARQUET", "1") != "0"
PLAIN_PARQUET_READ = os.environ.get("NVT_PLAIN_PARQUET_READ", "1") != "0"
PLAIN_READ_THREADS = int(os.environ.get("NVT_PARQUET_READ_THREADS", "32"))
_PLAIN_READ_POOL = None


def _plain_read_pool():
    global _PLAIN_READ_POOL
    if _PLAIN_READ_POOL is None:
        from concurrent.futures import ThreadPoolExecutor

        _PLAIN_READ_POOL = ThreadPoolExecutor(max_workers=PLAIN_READ_THREADS, thread_name_prefix="nvt-pqread")
    return _PLAIN_READ_POOL


class StagedPartition:
    """A partition as the hand-written parquet reader leaves it on the host: per column the
    packed non-null values and the validity bitmap in pinned memory
    (parquet_plain.read_row_groups_staged).  ``to_device`` enqueues the copies on the CURRENT
    stream (the prefetcher's side stream) and expands columns with nulls to one slot per row on
    the device (nvt_expand_valid)."""

    def __init__(self, columns):
        self.columns = columns

    @property
    def num_rows(self):
        return next(iter(self.columns.values())).rows if self.columns else 0

    def to_device(self, device=None):
        import ctypes as C

        import torch

        from . import kernels as K
        from .device import DeviceColumn, DeviceFrame, default_device

        device = device or default_device()
        out = {}
        for name, sc in self.columns.items():
            tdt = sc.values.dtype
            if device.type != "cuda":   # (host-only use: tests of the reader itself)
                raise K._lib.NvtHipError("StagedPartition.to_device needs a GPU")
            packed = sc.values[:sc.nvalid].to(device, non_blocking=True)
            if sc.valid is None:
                out[name] = DeviceColumn(packed)
                continue
            nb = ((sc.rows + 63) // 64) * 8
            bitmap = sc.valid[:nb].to(device, non_blocking=True)
            if sc.nvalid == 0:
                data = torch.zeros(sc.rows, dtype=tdt, device=device)
            else:
                data = torch.empty(sc.rows, dtype=tdt, device=device)
                need = C.c_uint64()
                lib = K._lib.load()
                K.check(lib.nvt_expand_valid_ws_bytes(sc.rows, C.byref(need)), "nvt_expand_valid_ws_bytes")
                ws = torch.empty(need.value, dtype=torch.uint8, device=device)
                K.check(lib.nvt_expand_valid(packed.data_ptr(), sc.dtype.itemsize, bitmap.data_ptr(), sc.rows,
                                             data.data_ptr(), ws.data_ptr(), K.stream_ptr()), "nvt_expand_valid")
            out[name] = DeviceColumn(data, bitmap)
        return DeviceFrame(out)
PLAIN_WRITE_THREADS = int(os.environ.get("NVT_PARQUET_THREADS", "16"))
PLAIN_ROW_GROUP = int(os.environ.get("NVT_PARQUET_ROW_GROUP", str(1 << 22)))
PLAIN_INFLIGHT = int(os.environ.get("NVT_PARQUET_INFLIGHT", "8"))   # row groups being written at once
LAST_TIMING = {}   # seconds of the last plain write: staging (enqueue + pinned allocation), waiting for copies, writing


def _plain_eligible(frame, dtypes) -> bool:
    """Every column a flat int32 / int64 / float32 / float64 device column (after the requested
    casts): the hand-written PLAIN writer takes the partition; anything else goes to pyarrow."""
    import numpy as np
    import torch

    from .parquet_plain import supported_dtype

    np_of = {torch.int32: "int32", torch.int64: "int64", torch.float32: "float32", torch.float64: "float64"}
    if len(frame.columns) == 0:
        return False
    for name, col in frame.items():
        if col.strings is not None or col.offsets is not None or col.data.dtype not in np_of:
            return False
        if dtypes and name in dtypes and not supported_dtype(np.dtype(dtypes[name])):
            return False
    return True


def _write_plain(parts, output_path, fname, k, shuffle, dtypes, statistics=False):
    """Dataset.to_parquet for fixed-width numeric frames: every partition is cut into row groups
    of PLAIN_ROW_GROUP rows; a row group's columns are copied into pinned host buffers on a side
    stream (nulls: values compacted and the validity bitmap re-packed on the device first) while
    the previous row group is written -- all its column chunks at once, by a pool of threads
    calling pwrite at offsets laid out beforehand (PlainParquetWriter).
    -> (names {file index: name}, rows {file index: rows}, file indices in order)."""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np
    import torch

    from . import kernels as K
    from .device import pack_bitmap_device
    from .parquet_plain import PlainParquetWriter

    import time

    t_of = {"int32": torch.int32, "int64": torch.int64, "float32": torch.float32, "float64": torch.float64}
    LAST_TIMING.update(wait_copy_s=0.0, write_s=0.0, stage_s=0.0, total_s=0.0, input_s=0.0, close_s=0.0)
    t_all = time.perf_counter()
    copy_s = None
    writers, names, rows_in = {}, {}, {}
    staged = deque()
    inflight = deque()   # (futures, host buffers) of row groups whose column writes are still running
    touched = set()

    with ThreadPoolExecutor(max_workers=PLAIN_WRITE_THREADS) as pool:
        def flush_one():
            j, cols, rows, event, keep, stats = staged.popleft()
            t0 = time.perf_counter()
            # the host does not wait for the copies: every column task synchronises with the
            # event itself before it writes.  Only validity bitmaps must be here already (the
            # pages are laid out from their popcounts).
            if event is not None and any(c[2] is not None for c in cols):
                event.synchronize()
            ready = event.synchronize if event is not None else None
            t1 = time.perf_counter()
            LAST_TIMING["wait_copy_s"] += t1 - t0
            w = writers.get(j)
            if w is None:
                names[j] = fname(j)
                w = writers[j] = PlainParquetWriter(
                    os.path.join(output_path, names[j]), [c[0] for c in cols],
                    [c[1].dtype for c in cols], pool=pool)
            elif w.names != [c[0] for c in cols] or w.dtypes != [c[1].dtype for c in cols]:
                # (pyarrow's ParquetWriter raises on a schema change too; never cast silently)
                raise ValueError(
                    f"to_parquet: partition schema {[(c[0], str(c[1].dtype)) for c in cols]} differs from "
                    f"the schema {list(zip(w.names, map(str, w.dtypes)))} of {names[j]}")
            # the column writes of this row group go to the pool and are NOT waited for: row
            # groups of other files (other inodes: buffered writes to ONE file serialise on its
            # inode lock, ~10 GB/s) and the next copies proceed meanwhile
            futs = w.write_row_group([(c[1], c[2]) for c in cols], rows, wait=False, ready=ready, stats=stats)
            inflight.append((futs, cols, keep))
            while len(inflight) > PLAIN_INFLIGHT:
                for f in inflight.popleft()[0]:
                    f.result()
            LAST_TIMING["write_s"] += time.perf_counter() - t1
            rows_in[j] = rows_in.get(j, 0) + rows

        parts = iter(parts)
        i = -1
        try:
            while True:
                t_in = time.perf_counter()
                part = next(parts, None)
                LAST_TIMING["input_s"] += time.perf_counter() - t_in
                if part is None:
                    break
                i += 1
                n = len(part)
                if shuffle is not None and n > 1:
                    part = part.take_rows(_device_permutation(n, part))
                cols = []
                for name, col in part.items():
                    col = col.materialize()
                    data = col.data
                    if dtypes and name in dtypes:
                        data = data.to(t_of[str(np.dtype(dtypes[name]))])
                    mask = K.unpack_bitmap(col.valid, n) if col.valid is not None else None
                    cols.append((name, data, mask))
                on_gpu = any(c[1].is_cuda for c in cols)
                if on_gpu and copy_s is None:
                    copy_s = torch.cuda.Stream()
                if on_gpu:
                    copy_s.wait_stream(torch.cuda.current_stream())
                pieces = [(i, 0, n)] if k is None else [
                    (j, (n * j) // k, (n * (j + 1)) // k) for j in range(k)]
                for j, a, b in pieces:
                    if b <= a and j in touched:
                        continue
                    touched.add(j)
                    for s0 in (range(a, b, PLAIN_ROW_GROUP) if b > a else [a]):
                        s1 = min(b, s0 + PLAIN_ROW_GROUP)
                        rows = s1 - s0
                        host, keep, stats = [], [], ([] if statistics else None)
                        t_st = time.perf_counter()
                        ctx = torch.cuda.stream(copy_s) if on_gpu else _nullcontext()
                        with ctx:
                            for name, data, mask in cols:
                                vals, bm = data[s0:s1], None
                                if mask is not None:
                                    m = mask[s0:s1]
                                    vals = vals[m]
                                    bm = pack_bitmap_device(m) if m.is_cuda else torch.from_numpy(
                                        np.packbits(m.numpy(), bitorder="little"))
                                hv = _to_host(vals)
                                hb = _to_host(bm) if bm is not None else None
                                if statistics:
                                    mm = None
                                    if vals.numel():
                                        if vals.dtype.is_floating_point:   # (NaN is no minimum / maximum)
                                            nan = torch.isnan(vals)
                                            lo = torch.where(nan, torch.full_like(vals, float("inf")), vals).amin()
                                            hi = torch.where(nan, torch.full_like(vals, float("-inf")), vals).amax()
                                        else:
                                            lo, hi = torch.aminmax(vals)
                                        mm = _to_host(torch.stack([lo, hi]))
                                        keep.append((mm,))
                                    stats.append(mm.numpy() if mm is not None else None)
                                keep.append((vals, bm))
                                host.append((name, hv.numpy(), hb.numpy() if hb is not None else None))
                            event = None
                            if on_gpu:
                                event = torch.cuda.Event()
                                event.record(copy_s)
                        LAST_TIMING["stage_s"] += time.perf_counter() - t_st
                        staged.append((j, host, rows, event, keep, stats))
                        flush_one()
            while staged:
                flush_one()
            t_cl = time.perf_counter()
            for w in writers.values():
                w.close()
            LAST_TIMING["close_s"] = time.perf_counter() - t_cl
        except BaseException:
            # no fds leaked, no truncated footer-less part files left behind
            for futs, _, _ in inflight:
                for f in futs:
                    try:
                        f.result()
                    except Exception:
                        pass
            for w in writers.values():
                w.abort()
            raise
    LAST_TIMING["total_s"] = time.perf_counter() - t_all
    return names, rows_in, sorted(writers)


class _nullcontext:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _to_host(t):
    import torch

    if not t.is_cuda:
        return t.contiguous()
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    return h


class Shuffle:
    """merlin.io.Shuffle: how ``Dataset.to_parquet`` randomises rows."""

    PER_PARTITION = "per-partition"
    PER_WORKER = "per-worker"
    FULL = "full"  # treated as PER_WORKER: there is one writer process per GPU

    @staticmethod
    def coerce(value):
        if value is None or value is False:
            return None
        if value is True:
            return Shuffle.PER_WORKER
        if value in (Shuffle.PER_PARTITION, Shuffle.PER_WORKER):
            return value
        if value == Shuffle.FULL:
            return Shuffle.PER_WORKER
        raise ValueError(f"unknown shuffle option {value!r}")


def _device_permutation(n: int, frame):
    import torch

    dev = None
    for _, col in frame.items():
        dev = col.data.device
        break
    return torch.randperm(n, device=dev)


def _prefetch_frames(host_parts, cols, depth: int = 2):
    import queue
    import threading

    import torch

    dev = torch.device("cuda", torch.cuda.current_device())
    side = torch.cuda.Stream(device=dev)
    q: "queue.Queue" = queue.Queue(maxsize=depth)
    _END = object()
    stop = threading.Event()  # set when the consumer abandons the generator early

    def put(item) -> bool:
        while not stop.is_set():
            try:
                q.put(item, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def producer():
        try:
            torch.cuda.set_device(dev)
            for part in host_parts:
                if stop.is_set():
                    return
                with torch.cuda.stream(side):
                    frame, _ = as_device_frame(part, dev)
                    if cols is not None:
                        frame = frame[[c for c in cols if c in frame]]
                    ev = torch.cuda.Event()
                    ev.record(side)
                if not put((frame, ev)):
                    return
            put(_END)
        except BaseException as e:  # surface decode / copy errors in the consumer
            put(e)

    t = threading.Thread(target=producer, daemon=True)
    t.start()
    try:
        while True:
            item = q.get()
            if item is _END:
                break
            if isinstance(item, BaseException):
                raise item
            frame, ev = item
            torch.cuda.current_stream().wait_event(ev)
            for _, col in frame.items():  # the buffers were allocated on the side stream
                for tns in (col.data, col.valid, col.offsets):
                    if tns is not None:
                        tns.record_stream(torch.cuda.current_stream())
            yield frame
    finally:
        # early exit (Dataset.head(), Workflow._capture_dtypes break after one partition, an
        # exception in the consumer): release the producer, drop the device-resident
        # partitions it still holds and join it -- it used to stay blocked in q.put forever
        stop.set()
        try:
            while True:
                q.get_nowait()
        except queue.Empty:
            pass
        t.join(timeout=30)


def _is_arrow_table(x) -> bool:
    try:
        import pyarrow as pa

        return isinstance(x, pa.Table)
    except ImportError:  # pragma: no cover
        return False


def _accepts_columns(fn) -> bool:
    try:
        from inspect import signature

        return "columns" in signature(fn).parameters
    except (TypeError, ValueError):
        return False


def _split(frame, n):
    if n <= 1:
        return [frame]
    total = len(frame)
    step = -(-total // n)
    step = -(-step // 8) * 8  # bitmap-friendly boundaries
    out = []
    for s in range(0, total, step):
        e = min(total, s + step)
        if isinstance(frame, pd.DataFrame):
            out.append(frame.iloc[s:e].reset_index(drop=True))
        elif isinstance(frame, DeviceFrame):
            out.append(frame.slice_rows(s, e))
        else:
            out.append(frame.slice(s, e - s))
    return out
