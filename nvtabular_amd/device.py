"""HBM-resident columnar frames.

The reference's operators receive pandas or cuDF dataframes
(nvtabular/ops/operator.py:24-27).  cuDF does not exist on ROCm, so this engine's
device-side frame is ``DeviceFrame``: an ordered dict of ``DeviceColumn`` whose
buffers live in HBM in Arrow layout --

  data     1-D contiguous values (int32/int64/float32/float64/uint8/bool)
  valid    optional Arrow validity bitmap (uint8, LSB first, 1 = valid)
  offsets  optional int64 row offsets (list columns; ``data`` holds the leaves)
  fill     optional pending FillMissing constant (logical value of a null row);
           downstream kernels take it as a parameter so FillMissing costs no pass

Operators accept either a pandas DataFrame (converted on entry and back on exit,
so existing workflows drop in) or a DeviceFrame (stays resident).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import os

import numpy as np
import pandas as pd
import torch

from . import kernels as K

_NP_TO_TORCH = {
    np.dtype("int32"): torch.int32,
    np.dtype("int64"): torch.int64,
    np.dtype("float32"): torch.float32,
    np.dtype("float64"): torch.float64,
    np.dtype("uint8"): torch.uint8,
    np.dtype("bool"): torch.bool,
}
_TORCH_TO_NP = {v: k for k, v in _NP_TO_TORCH.items()}


def torch_dtype(dt) -> torch.dtype:
    if isinstance(dt, torch.dtype):
        return dt
    return _NP_TO_TORCH[np.dtype(dt)]


def numpy_dtype(dt: torch.dtype) -> np.dtype:
    return _TORCH_TO_NP[dt]


def default_device() -> torch.device:
    K._lib.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def to_device_async(arr: np.ndarray, device) -> torch.Tensor:
    """Host array -> HBM: ONE host copy into a pinned staging tensor (torch's caching host
    allocator recycles the blocks), then an asynchronous copy on the CURRENT stream (the
    parquet prefetcher makes that a side stream, so it overlaps the kernels of the previous
    partition; hipMemcpyAsync is only asynchronous from pinned memory).  Read-only Arrow
    buffers are fine: they are only read."""
    arr = np.ascontiguousarray(arr)
    if device.type != "cuda" or arr.dtype not in _NP_TO_TORCH:
        return torch.from_numpy(arr.copy() if not arr.flags.writeable else arr).to(device)
    pin = torch.empty(arr.shape, dtype=_NP_TO_TORCH[arr.dtype], pin_memory=True)
    np.copyto(pin.numpy(), arr)  # releases the GIL: columns are staged from several threads
    return pin.to(device, non_blocking=True)


_STAGE_POOL = None


def _stage_pool():
    global _STAGE_POOL
    if _STAGE_POOL is None:
        from concurrent.futures import ThreadPoolExecutor

        _STAGE_POOL = ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1),
                                         thread_name_prefix="nvt-stage")
    return _STAGE_POOL


def pack_bitmap(valid_bool: np.ndarray) -> np.ndarray:
    """bool[n] -> Arrow LSB-first bitmap, padded to a multiple of 8 bytes."""
    bits = np.packbits(valid_bool.astype(np.uint8), bitorder="little")
    pad = (-len(bits)) % 8
    if pad:
        bits = np.concatenate([bits, np.zeros(pad, dtype=np.uint8)])
    return bits


class DeviceColumn:
    __slots__ = ("data", "valid", "offsets", "fill", "strings")

    def __init__(self, data, valid=None, offsets=None, fill=None, strings=None):
        self.data = data
        self.valid = valid
        self.offsets = offsets
        self.fill = fill
        self.strings = strings  # host dict {surrogate int64 key -> str} for string columns

    # ---- basic properties -------------------------------------------------
    def __len__(self):
        return int(self.offsets.numel() - 1) if self.offsets is not None else int(self.data.numel())

    @property
    def dtype(self) -> torch.dtype:
        return self.data.dtype

    @property
    def is_list(self) -> bool:
        return self.offsets is not None

    @property
    def device(self):
        return self.data.device

    def shallow_copy(self) -> "DeviceColumn":
        return DeviceColumn(self.data, self.valid, self.offsets, self.fill, self.strings)

    def with_data(self, data, valid=None, keep_offsets=True) -> "DeviceColumn":
        return DeviceColumn(data, valid, self.offsets if keep_offsets else None, None, None)

    def null_count(self) -> int:
        n = int(self.data.numel())
        nulls = n - K.popcount(self.valid, n)
        if self.data.dtype in (torch.float32, torch.float64):
            nulls += int(torch.isnan(self.data).sum().item())
        return nulls

    def materialize(self) -> "DeviceColumn":
        """Apply a pending FillMissing constant (one fused fill pass)."""
        if self.fill is None:
            return self
        data = self.data
        if data.dtype == torch.bool:
            data = data.view(torch.uint8)
        out_dt = data.dtype
        fv = float(self.fill)
        if out_dt in (torch.int32, torch.int64) and fv != int(fv):
            out_dt = torch.float64
        if out_dt == torch.uint8:
            out_dt = torch.float64
        out, _ = K.fill_normalize(data, self.valid, fv, False, 0.0, 1.0, out_dt)
        return DeviceColumn(out, None, self.offsets, None, self.strings)

    # ---- host conversion --------------------------------------------------
    @staticmethod
    def from_pandas(s: pd.Series, device=None) -> "DeviceColumn":
        device = device or default_device()
        from .strings import string_column_to_device  # local: avoids a cycle

        if s.dtype == object or pd.api.types.is_string_dtype(s.dtype):
            nn = s.dropna()
            if len(nn) and isinstance(nn.iloc[0], (list, np.ndarray, tuple)):
                lens = np.fromiter((len(r) if r is not None else 0 for r in s), dtype=np.int64,
                                   count=len(s))
                offsets = np.zeros(len(s) + 1, dtype=np.int64)
                np.cumsum(lens, out=offsets[1:])
                leaves = [v for r in s if r is not None for v in r]
                leaf_col = DeviceColumn.from_pandas(pd.Series(leaves), device)
                if len(leaves) == 0:
                    leaf_col = DeviceColumn(torch.empty(0, dtype=torch.int64, device=device))
                leaf_col.offsets = torch.from_numpy(offsets).to(device)
                return leaf_col
            return string_column_to_device(s, device)
        if isinstance(s.dtype, pd.api.extensions.ExtensionDtype):
            # pandas nullable / arrow-backed numerics: values + mask
            mask = s.isna().to_numpy()
            np_dt = getattr(s.dtype, "numpy_dtype", None)
            if np_dt is None:
                np_dt = np.dtype(s.dtype.pyarrow_dtype.to_pandas_dtype())
            vals = s.fillna(0).to_numpy(dtype=np_dt)
            data = torch.from_numpy(np.ascontiguousarray(vals)).to(device)
            valid = torch.from_numpy(pack_bitmap(~mask)).to(device) if mask.any() else None
            return DeviceColumn(data, valid)
        arr = np.ascontiguousarray(s.to_numpy())
        if arr.dtype not in _NP_TO_TORCH:
            if arr.dtype.kind in "iu":
                arr = arr.astype(np.int64)
            elif arr.dtype.kind == "f":
                arr = arr.astype(np.float64)
            else:
                raise TypeError(f"unsupported column dtype {arr.dtype}")
        return DeviceColumn(torch.from_numpy(arr).to(device))

    @staticmethod
    def from_arrow(arr, device=None) -> "DeviceColumn":
        """pyarrow Array/ChunkedArray -> device, reusing Arrow's buffers (no pandas)."""
        import pyarrow as pa

        device = device or default_device()
        if isinstance(arr, pa.ChunkedArray):
            arr = arr.combine_chunks() if arr.num_chunks != 1 else arr.chunk(0)
        if pa.types.is_list(arr.type) or pa.types.is_large_list(arr.type):
            leaves = DeviceColumn.from_arrow(arr.flatten(), device)
            off = np.asarray(arr.offsets).astype(np.int64)
            leaves.offsets = torch.from_numpy(off - off[0]).to(device)
            return leaves
        if pa.types.is_string(arr.type) or pa.types.is_large_string(arr.type) or \
                pa.types.is_dictionary(arr.type):
            return DeviceColumn.from_pandas(arr.to_pandas(), device)
        np_dt = arr.type.to_pandas_dtype()
        n = len(arr)
        valid = None
        if arr.null_count:
            # Arrow already stores validity as an LSB-first bitmap: reuse its buffer when the
            # array is not sliced, otherwise re-pack
            bufs = arr.buffers()
            if arr.offset == 0 and bufs[0] is not None:
                bits = np.frombuffer(bufs[0], dtype=np.uint8)[: (n + 7) // 8]
                pad = (-len(bits)) % 8
                if pad:
                    bits = np.concatenate([bits, np.zeros(pad, dtype=np.uint8)])
                valid = to_device_async(bits, device)
            else:
                valid = to_device_async(pack_bitmap(np.asarray(arr.is_valid())), device)
            if pa.types.is_boolean(arr.type) or bufs[1] is None:
                vals = arr.fill_null(False).to_numpy(zero_copy_only=False).astype(np_dt, copy=False)
            else:
                # the values buffer as it is: slots under a null hold arbitrary bytes, every
                # kernel goes by the bitmap (a fill_null(0) here was a full extra host copy)
                vals = np.frombuffer(bufs[1], dtype=np.dtype(np_dt), count=n,
                                     offset=arr.offset * np.dtype(np_dt).itemsize)
        else:
            vals = arr.to_numpy(zero_copy_only=False)
        assert len(vals) == n
        if vals.dtype not in _NP_TO_TORCH:
            vals = vals.astype(np.int64 if vals.dtype.kind in "iu" else np.float64)
        data = to_device_async(vals, device)
        return DeviceColumn(data, valid)

    def valid_mask_host(self) -> Optional[np.ndarray]:
        if self.valid is None:
            return None
        n = int(self.data.numel())
        bits = np.unpackbits(self.valid.cpu().numpy(), bitorder="little")[:n]
        return bits.astype(bool)

    def to_pandas(self, name=None) -> pd.Series:
        col = self.materialize()
        vals = col.data.cpu().numpy()
        mask = col.valid_mask_host()
        if col.strings is not None:
            lut = col.strings
            out = np.array([lut.get(int(k)) for k in vals], dtype=object)
            if mask is not None:
                out[~mask] = None
            flat = pd.Series(out, name=name)
        elif mask is not None and not mask.all():
            if vals.dtype.kind in "iub":
                vals = vals.astype(np.float64)  # pandas' int-with-null convention
            else:
                vals = vals.copy()
            vals[~mask] = np.nan
            flat = pd.Series(vals, name=name)
        else:
            flat = pd.Series(vals, name=name)
        if col.offsets is None:
            return flat
        off = col.offsets.cpu().numpy()
        fv = flat.to_numpy()
        rows = [fv[off[i] : off[i + 1]] for i in range(len(off) - 1)]
        return pd.Series(rows, name=name, dtype=object)


class DeviceFrame:
    """Ordered mapping name -> DeviceColumn with a common row count."""

    def __init__(self, columns: Optional[Dict[str, DeviceColumn]] = None):
        self._cols: Dict[str, DeviceColumn] = dict(columns or {})

    # ---- dataframe-ish surface used by the operators ------------------------
    @property
    def columns(self) -> List[str]:
        return list(self._cols)

    def __contains__(self, name):
        return name in self._cols

    def __len__(self):
        for c in self._cols.values():
            return len(c)
        return 0

    def __getitem__(self, key):
        if isinstance(key, (list, tuple)):
            return DeviceFrame({k: self._cols[k] for k in key})
        return self._cols[key]

    def __setitem__(self, name, col):
        if not isinstance(col, DeviceColumn):
            raise TypeError("DeviceFrame values must be DeviceColumn")
        self._cols[name] = col

    def __iter__(self):
        return iter(self._cols)

    def items(self):
        return self._cols.items()

    def copy(self, deep=False) -> "DeviceFrame":
        return DeviceFrame({k: v.shallow_copy() for k, v in self._cols.items()})

    def take_rows(self, index: torch.Tensor) -> "DeviceFrame":
        """Rows in the order of ``index`` (int64 on the device) -- the per-partition shuffle
        of Dataset.to_parquet.  Torch gathers: output plumbing, not on the fit/transform path."""
        out = DeviceFrame()
        for name, col in self._cols.items():
            col = col.materialize()
            if col.is_list:
                starts = col.offsets[:-1][index]
                lens = (col.offsets[1:] - col.offsets[:-1])[index]
                new_off = torch.zeros(index.numel() + 1, dtype=torch.int64, device=index.device)
                torch.cumsum(lens, 0, out=new_off[1:])
                total = int(new_off[-1].item())
                within = torch.arange(total, device=index.device) - torch.repeat_interleave(new_off[:-1], lens)
                leaf = torch.repeat_interleave(starts, lens) + within
                take, offsets = leaf, new_off
            else:
                take, offsets = index, None
            valid = None
            if col.valid is not None:
                bits = (col.valid[take >> 3] >> (take & 7).to(torch.uint8)) & 1
                valid = pack_bitmap_device(bits.to(torch.bool))
            out[name] = DeviceColumn(col.data[take], valid, offsets, None, col.strings)
        return out

    def drop(self, columns: Iterable[str]) -> "DeviceFrame":
        drop = set(columns)
        return DeviceFrame({k: v for k, v in self._cols.items() if k not in drop})

    def rename(self, mapping: Dict[str, str]) -> "DeviceFrame":
        return DeviceFrame({mapping.get(k, k): v for k, v in self._cols.items()})

    @staticmethod
    def concat_columns(frames: Iterable["DeviceFrame"]) -> "DeviceFrame":
        out = DeviceFrame()
        for f in frames:
            if f is None:
                continue
            for k, v in f.items():
                out._cols[k] = v
        return out

    # ---- conversions ----------------------------------------------------------
    @staticmethod
    def from_pandas(df: pd.DataFrame, device=None) -> "DeviceFrame":
        device = device or default_device()
        return DeviceFrame({c: DeviceColumn.from_pandas(df[c], device) for c in df.columns})

    @staticmethod
    def from_arrow(table, device=None) -> "DeviceFrame":
        """Columns are staged (host memcpy into pinned memory) by a small thread pool -- one
        thread copies at ~5 GB/s, a fifth of what the PCIe link takes -- and every copy is
        enqueued on the caller's current stream."""
        device = device or default_device()
        names = table.column_names
        if device.type != "cuda" or len(names) < 2 or table.num_rows < (1 << 16):
            return DeviceFrame({n: DeviceColumn.from_arrow(table.column(n), device) for n in names})
        stream = torch.cuda.current_stream(device)

        def stage(name):
            torch.cuda.set_device(device)
            with torch.cuda.stream(stream):
                return DeviceColumn.from_arrow(table.column(name), device)

        cols = list(_stage_pool().map(stage, names))
        return DeviceFrame(dict(zip(names, cols)))

    def to_pandas(self) -> pd.DataFrame:
        return pd.DataFrame({k: v.to_pandas(k) for k, v in self._cols.items()})

    def to_arrow(self):
        """HBM -> pyarrow.Table without going through pandas: every buffer (values, validity
        bitmap, list offsets) is copied into pinned host memory with asynchronous copies, ONE
        stream synchronisation, and wrapped as Arrow arrays in place -- the output half of the
        parquet path (Dataset.to_parquet).  Integer columns keep their type and their nulls
        (pandas would turn them into float64); string columns are decoded on the host."""
        import pyarrow as pa

        staged = {}
        for name, col in self._cols.items():
            col = col.materialize()
            if col.strings is not None or not col.data.is_cuda:
                staged[name] = ("pandas", col)
                continue
            data = col.data.view(torch.uint8) if col.data.dtype == torch.bool else col.data

            def pin(t):
                if t is None:
                    return None
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                h.copy_(t, non_blocking=True)
                return h

            staged[name] = ("arrow", col, pin(data.contiguous()), pin(col.valid), pin(col.offsets))
        if any(v[0] == "arrow" for v in staged.values()):
            torch.cuda.current_stream().synchronize()
        arrays = {}
        for name, item in staged.items():
            if item[0] == "pandas":
                arrays[name] = pa.Array.from_pandas(item[1].to_pandas(name))
                continue
            _, col, hdata, hvalid, hoff = item
            n_leaf = int(hdata.numel())
            values = hdata.numpy()
            if col.data.dtype == torch.bool:
                arr = pa.array(values.astype(bool))
                if hvalid is not None:
                    mask = np.unpackbits(hvalid.numpy(), bitorder="little")[:n_leaf] == 0
                    arr = pa.array(values.astype(bool), mask=mask)
            else:
                bufs = [None, pa.py_buffer(values)]
                nulls = 0
                if hvalid is not None:
                    vb = hvalid.numpy()
                    bufs[0] = pa.py_buffer(vb)
                    nulls = -1  # let Arrow count
                arr = pa.Array.from_buffers(pa.from_numpy_dtype(values.dtype), n_leaf, bufs,
                                            null_count=nulls)
            if hoff is not None:
                arr = pa.LargeListArray.from_arrays(pa.array(hoff.numpy()), arr)
            arrays[name] = arr
        return pa.table(arrays)

    def slice_rows(self, start: int, stop: int) -> "DeviceFrame":
        out = DeviceFrame()
        for k, c in self._cols.items():
            if c.is_list:
                raise NotImplementedError("row slicing of list columns")
            valid = None
            if c.valid is not None:
                if start % 8:
                    raise ValueError("partition starts must be multiples of 8 rows for bitmaps")
                valid = c.valid[start // 8 :]
            out[k] = DeviceColumn(c.data[start:stop], valid, None, c.fill, c.strings)
        return out


def as_device_frame(df, device=None):
    """(DeviceFrame, was_pandas)"""
    if isinstance(df, DeviceFrame):
        return df, False
    if isinstance(df, pd.DataFrame):
        return DeviceFrame.from_pandas(df, device), True
    if hasattr(df, "to_device") and type(df).__name__ == "StagedPartition":
        return df.to_device(device), False   # (io.StagedPartition: the hand-written parquet reader)
    try:
        import pyarrow as pa

        if isinstance(df, pa.Table):
            return DeviceFrame.from_arrow(df, device), False
    except ImportError:  # pragma: no cover
        pass
    raise TypeError(f"unsupported frame type {type(df)}")


def pack_bitmap_device(mask: torch.Tensor) -> torch.Tensor:
    """bool[n] on device -> Arrow LSB-first bitmap (torch plumbing; rare paths only)."""
    n = mask.numel()
    pad = (-n) % 64
    m = mask.to(torch.uint8)
    if pad:
        m = torch.cat([m, torch.zeros(pad, dtype=torch.uint8, device=mask.device)])
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=mask.device)
    return (m.view(-1, 8) * w).sum(dim=1, dtype=torch.int32).to(torch.uint8)


def key_view(col: "DeviceColumn"):
    """(int keys tensor, validity bitmap) of a categorical column's values/leaves.
    Float key columns (pandas' int-with-null artefact) are cast to int64 with
    NaN -> null."""
    data, valid = col.data, col.valid
    if data.dtype in (torch.int32, torch.int64):
        return data, valid
    if data.dtype in (torch.uint8, torch.bool):
        return K.widen_i64(data), valid
    if data.dtype in (torch.float32, torch.float64):
        nan = torch.isnan(data)
        keys = torch.where(nan, torch.zeros_like(data), data).to(torch.int64)
        if bool(nan.any()):
            m = ~nan
            if valid is not None:
                host = col.valid_mask_host()
                m = m & torch.from_numpy(host).to(data.device)
            valid = pack_bitmap_device(m)
        return keys, valid
    raise TypeError(f"unsupported categorical dtype {data.dtype}")
