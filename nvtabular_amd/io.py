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

        def gen(columns=None):
            for f, groups in pieces:
                yield pq.ParquetFile(f).read_row_groups(groups, columns=columns)

        self._parts_fn = gen

    # ---- public surface -----------------------------------------------------------
    @property
    def schema(self) -> Schema:
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
        it = self._parts_fn(cols) if _accepts_columns(self._parts_fn) else self._parts_fn()
        for i, part in enumerate(it):
            if shard is not None and i % shard[1] != shard[0]:
                continue
            yield part

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
                   cats=None, conts=None, labels=None, **_):
        """One parquet file per partition under output_path (plus nothing else):
        the I/O path is SURVEY section 8(f) item 2, kept minimal here."""
        import pyarrow as pa
        import pyarrow.parquet as pq

        os.makedirs(output_path, exist_ok=True)
        for i, part in enumerate(self.to_iter()):
            df = part.to_pandas()
            if dtypes:
                df = df.astype({k: v for k, v in dtypes.items() if k in df.columns})
            pq.write_table(pa.Table.from_pandas(df, preserve_index=False),
                           os.path.join(output_path, f"part_{i}.parquet"))


def _prefetch_frames(host_parts, cols, depth: int = 2):
    import queue
    import threading

    import torch

    dev = torch.device("cuda", torch.cuda.current_device())
    side = torch.cuda.Stream(device=dev)
    q: "queue.Queue" = queue.Queue(maxsize=depth)
    _END = object()

    def producer():
        try:
            torch.cuda.set_device(dev)
            for part in host_parts:
                with torch.cuda.stream(side):
                    frame, _ = as_device_frame(part, dev)
                    if cols is not None:
                        frame = frame[[c for c in cols if c in frame]]
                    ev = torch.cuda.Event()
                    ev.record(side)
                q.put((frame, ev))
            q.put(_END)
        except BaseException as e:  # surface decode / copy errors in the consumer
            q.put(e)

    t = threading.Thread(target=producer, daemon=True)
    t.start()
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
    t.join()


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
