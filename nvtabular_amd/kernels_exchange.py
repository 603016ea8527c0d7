"""Device work around the multi-GPU exchange of (key, count) rows (dist.merge_counts_many):
nvt_exchange_* and the owner-side sorted merge.

Part of the host driver of the C ABI (include/nvt_hip.h); ``kernels.py`` is the facade every
caller imports -- it re-exports these names, holds the run-time switches they read (``K.<FLAG>`` at
call time: tests and A / B runs set them on the facade) and the helpers they share."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from . import kernels as K
from ._lib import check


class ExchangeBatch:
    """The (keys int32, counts int64) lists of ALL columns of a fit as one descriptor array for
    the nvt_exchange_* launches of dist.merge_counts_many."""

    def __init__(self, tables):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.keep = [(k.contiguous(), c.contiguous()) for k, c in tables]
        self.ncol = len(self.keep)
        self.total = sum(int(k.numel()) for k, _ in self.keep)
        self.dev = self.keep[0][0].device
        self.cols = (_lib.XCol * self.ncol)()
        for j, (k, c) in enumerate(self.keep):
            assert k.dtype == torch.int32 and c.dtype == torch.int64
            self.cols[j].keys, self.cols[j].counts, self.cols[j].n = K.ptr(k) or 0, K.ptr(c) or 0, int(k.numel())

    def ranges(self) -> torch.Tensor:
        """int64[ncol, 3] = (-min key, max key, sum of counts); a column without entries:
        (-INT64_MAX, -INT64_MAX, 0)."""
        rng = torch.empty((self.ncol, 3), dtype=torch.int64, device=self.dev)
        check(self.lib.nvt_exchange_ranges(self.cols, self.ncol, rng.data_ptr(), K.stream_ptr()),
              "nvt_exchange_ranges")
        return rng

    def ranges_sorted(self) -> torch.Tensor:
        """The same for KEY-SORTED lists: (-first key, last key, 0) -- no pass over the lists."""
        rng = torch.empty((self.ncol, 3), dtype=torch.int64, device=self.dev)
        check(self.lib.nvt_exchange_ranges_sorted(self.cols, self.ncol, rng.data_ptr(), K.stream_ptr()),
              "nvt_exchange_ranges_sorted")
        return rng

    def _owner_args(self, lo, width):
        return ((C.c_int64 * self.ncol)(*[int(v) for v in lo]),
                (C.c_uint64 * self.ncol)(*[max(1, int(v)) for v in width]))

    def hist(self, lo, width, G) -> torch.Tensor:
        """int64[G, ncol]: rows of column j whose key range belongs to rank g."""
        mat = torch.empty((G, self.ncol), dtype=torch.int64, device=self.dev)
        a, b = self._owner_args(lo, width)
        check(self.lib.nvt_exchange_hist(self.cols, self.ncol, a, b, G, mat.data_ptr(), K.stream_ptr()),
              "nvt_exchange_hist")
        return mat

    def scatter(self, lo, width, G, starts: torch.Tensor) -> torch.Tensor:
        """(count << 32 | key) words grouped by (owner, column); ``starts`` int64[G, ncol] (device)
        = the first position of every group (consumed: advanced to the group ends)."""
        rows = torch.empty(self.total, dtype=torch.int64, device=self.dev)
        if self.total == 0:
            return rows  # (a rank that received no partition: nothing to send, nothing to launch)
        a, b = self._owner_args(lo, width)
        check(self.lib.nvt_exchange_scatter(self.cols, self.ncol, a, b, G, starts.data_ptr(),
                                            rows.data_ptr(), K.stream_ptr()), "nvt_exchange_scatter")
        return rows


    def pack_ordered(self, lo, width, G, starts: torch.Tensor, first_row: torch.Tensor) -> torch.Tensor:
        """The send buffer of KEY-SORTED lists: group (g, j) = the contiguous slice of column j
        whose keys rank g owns, copied in key order to ``starts[g, j]``; ``first_row[g, j]`` =
        rows of column j in front of the slice (both int64[G, ncol] on the device).  No atomics."""
        rows = torch.empty(self.total, dtype=torch.int64, device=self.dev)
        if self.total == 0:
            return rows
        a, b = self._owner_args(lo, width)
        check(self.lib.nvt_exchange_pack_ordered(self.cols, self.ncol, a, b, G, first_row.data_ptr(),
                                                 starts.data_ptr(), rows.data_ptr(), K.stream_ptr()),
              "nvt_exchange_pack_ordered")
        return rows


def exchange_unpack(words: torch.Tensor, seg_off: List[int], dst_off: List[int], out_n: int, extra=None):
    """Gathered (count << 32 | key) words in segments -> (keys int32[out_n], counts int64[out_n]),
    segment s copied to position dst_off[s] (column-major: every column one contiguous list).
    extra: one int32 per word that travels along -> a third result (int32[out_n])."""
    _lib.require_gpu()
    n = int(words.numel())
    dev = words.device
    keys = torch.empty(out_n, dtype=torch.int32, device=dev)
    cnts = torch.empty(out_n, dtype=torch.int64, device=dev)
    so = torch.tensor(seg_off, dtype=torch.int64, device=dev)
    do = torch.tensor(dst_off, dtype=torch.int64, device=dev)
    if extra is not None:
        assert extra.dtype == torch.int32 and int(extra.numel()) == n
        xo = torch.empty(out_n, dtype=torch.int32, device=dev)
        check(_lib.load().nvt_exchange_unpack2(words.contiguous().data_ptr(), extra.contiguous().data_ptr(), n,
                                               so.data_ptr(), do.data_ptr(), len(seg_off) - 1, keys.data_ptr(),
                                               cnts.data_ptr(), xo.data_ptr(), K.stream_ptr()),
              "nvt_exchange_unpack2")
        return keys, cnts, xo
    check(_lib.load().nvt_exchange_unpack(words.contiguous().data_ptr(), n, so.data_ptr(), do.data_ptr(),
                                          len(seg_off) - 1, keys.data_ptr(), cnts.data_ptr(),
                                          K.stream_ptr()), "nvt_exchange_unpack")
    return keys, cnts


def merge_counts_sorted(rows: torch.Tensor, seg_off: List[int], ncol: int, want_packed=False):
    """nvt_count_merge_sorted: owner-side merge of received (count << 32 | int32 key) rows lying in
    len(seg_off) - 1 segments (source-major, column-minor).  Returns [(keys int32, counts
    int64)] per column, every list ordered by key.  One read-back (groups per column).
    want_packed: also the merged rows of all columns as ONE (count << 32 | key) array, column
    after column (what the all-gather sends), and the per-column lengths."""
    _lib.require_gpu()
    lib = _lib.load()
    dev = rows.device
    n = int(rows.numel())
    empty = lambda: (torch.empty(0, dtype=torch.int32, device=dev),  # noqa: E731
                     torch.empty(0, dtype=torch.int64, device=dev))
    if n == 0:
        out = [empty() for _ in range(ncol)]
        return (out, torch.empty(0, dtype=torch.int64, device=dev), [0] * ncol) if want_packed else out
    rows = rows.contiguous()
    off = torch.tensor(seg_off, dtype=torch.int64, device=dev)
    need = C.c_uint64()
    check(lib.nvt_count_merge_sorted_ws_bytes(n, C.byref(need)), "nvt_count_merge_sorted_ws_bytes")
    ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    keys = torch.empty(n, dtype=torch.int32, device=dev)
    col = torch.empty(n, dtype=torch.int64, device=dev)
    sums = torch.empty(n, dtype=torch.float64, device=dev)
    state = torch.empty(_lib.STATE_WORDS, dtype=torch.int64, device=dev)
    check(lib.nvt_count_merge_sorted(rows.data_ptr(), n, off.data_ptr(), len(seg_off) - 1, ncol,
                                     keys.data_ptr(), col.data_ptr(), sums.data_ptr(),
                                     state.data_ptr(), ws.data_ptr(), K.stream_ptr()),
          "nvt_count_merge_sorted")
    g = int(K.read_back(state)[_lib.ST_OCCUPIED])
    # groups are ordered by (column, key): the first group of column j = groups of smaller columns
    bounds = torch.searchsorted(col[:g], torch.arange(ncol + 1, dtype=torch.int64, device=dev))
    bounds = K.read_back(bounds.to(torch.int64)).tolist()
    counts = sums[:g].to(torch.int64)
    out = []
    for j in range(ncol):
        lo, hi = int(bounds[j]), int(bounds[j + 1])
        out.append((keys[lo:hi], counts[lo:hi]) if hi > lo else empty())
    if want_packed:
        packed = (counts << 32) | (keys[:g].to(torch.int64) & 0xFFFFFFFF)
        return out, packed, [int(bounds[j + 1]) - int(bounds[j]) for j in range(ncol)]
    return out
