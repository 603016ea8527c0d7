"""Thin, typed wrappers over the C ABI: torch tensors in, torch tensors out.

torch is used here only as the device allocator / stream provider ("plumbing");
every O(rows) computation below is a hand-written gfx950 kernel reached through
``libnvt_hip.so``.  Nothing in this module has a CPU code path.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import math
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import check

INT32_MIN = -(2**31)
INT64_MIN = -(2**63)

_DTYPE_CODE = {
    torch.float32: _lib.NVT_F32,
    torch.float64: _lib.NVT_F64,
    torch.int32: _lib.NVT_I32,
    torch.int64: _lib.NVT_I64,
    torch.uint8: _lib.NVT_U8,
    torch.bool: _lib.NVT_U8,
}


# --------------------------------------------------------------------------
# per-kernel timing (bench.py): HIP events recorded INSIDE the library, on the stream each
# kernel family is launched on (nvt_prof_begin / nvt_prof_report); nothing is recorded -- and
# nothing is paid -- while profiling is off
# --------------------------------------------------------------------------
def profile_begin():
    check(_lib.load().nvt_prof_begin(), "nvt_prof_begin")


def profile_report():
    """dict(kernels={name: (total_ms, launches, algorithmic_bytes)}, busy_ms, span_ms) for the
    profiled region; busy_ms = union of the recorded kernel intervals (GPU-busy time)."""
    import json

    lib = _lib.load()
    need = C.c_uint64()
    buf = C.create_string_buffer(1 << 16)
    check(lib.nvt_prof_report(buf, len(buf), C.byref(need)), "nvt_prof_report")
    rep = json.loads(buf.value.decode())
    rep["kernels"] = {k: tuple(v) for k, v in rep["kernels"].items()}
    return rep


def profile_end():
    """{kernel: (total_ms, launches, algorithmic_bytes)} for the profiled region."""
    return profile_report()["kernels"]


class annotate:
    """roctx range named after the reference's @annotate of the step being replaced
    (categorify.py:345,477,955,1054,1073,1149): attributes rocprofv3 traces to operators."""

    def __init__(self, name: str):
        self.name = name.encode()

    def __enter__(self):
        try:
            _lib.load().nvt_range_push(self.name)
            self._on = True
        except Exception:
            self._on = False
        return self

    def __exit__(self, *exc):
        if self._on:
            _lib.load().nvt_range_pop()
        return False


_event_pool = []  # handles of released Events: a fit re-uses them instead of creating new ones


class Event:
    """Opaque nvt_event (a HIP event without timing): recorded by the library behind the last
    kernel that produces an object on one of its internal streams, waited for -- on the stream,
    never on the host -- by whoever consumes the object next.  Handles are pooled: a steady-state
    fit creates and destroys no HIP objects."""

    def __init__(self):
        if _event_pool:
            self.handle = _event_pool.pop()
            return
        h = C.c_void_p()
        check(_lib.load().nvt_event_create(C.byref(h)), "nvt_event_create")
        self.handle = h

    def wait(self, stream: Optional[int] = None):
        check(_lib.load().nvt_stream_wait_event(stream_ptr() if stream is None else stream,
                                                self.handle), "nvt_stream_wait_event")

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self.handle = None
            try:
                if len(_event_pool) < 256:
                    _event_pool.append(h)  # waits already enqueued captured the event's state
                else:
                    _lib.load().nvt_event_destroy(h)
            except Exception:
                pass


_mailboxes = {}
READBACK_TIMEOUT_S = float(os.environ.get("NVT_READBACK_TIMEOUT", "120"))


READBACK_MEMCPY = os.environ.get("NVT_READBACK", "mailbox") == "memcpy"   # (read once, at import)


def read_back(t: torch.Tensor):
    """Small device tensor (int64 / float64) -> numpy array on the host WITHOUT a blocking
    runtime wait: one tiny kernel copies it into coherent pinned memory and the host spins on
    a sequence word (nvt_mailbox_*; include/nvt_hip.h explains why).  Stream-ordered behind
    everything queued on the current stream, like ``t.cpu()``."""
    import numpy as np

    assert t.is_cuda and t.element_size() == 8
    if READBACK_MEMCPY:
        return t.cpu().numpy()
    lib = _lib.load()
    t = t.contiguous()
    nbytes = t.numel() * 8
    key = (t.device.index, stream_ptr())
    mb = _mailboxes.get(key)
    if mb is None or lib.nvt_mailbox_capacity(mb) < nbytes:
        if mb is not None:
            lib.nvt_mailbox_destroy(mb)
        h = C.c_void_p()
        check(lib.nvt_mailbox_create(max(1 << 16, 2 * nbytes), C.byref(h)), "nvt_mailbox_create")
        mb = _mailboxes[key] = h
    if nbytes == 0:
        return np.empty(t.shape, dtype=np.int64 if t.dtype == torch.int64 else np.float64)
    seq = C.c_uint64()
    check(lib.nvt_mailbox_post(mb, t.data_ptr(), nbytes, stream_ptr(), C.byref(seq)),
          "nvt_mailbox_post")
    check(lib.nvt_mailbox_wait(mb, seq.value, READBACK_TIMEOUT_S), "nvt_mailbox_wait")
    np_dt = {torch.int64: np.int64, torch.float64: np.float64}[t.dtype]
    buf = (C.c_char * nbytes).from_address(lib.nvt_mailbox_data(mb))
    return np.frombuffer(buf, dtype=np_dt).reshape(tuple(t.shape)).copy()


class PendingReadBack:
    """read_back in two halves: the copy is enqueued by the constructor, ``get()`` waits for it.
    Between the two the host can keep enqueueing work BEHIND the copy: by the time it asks, the
    value has arrived and the device is still busy (a blocking read-back leaves the device idle
    for as long as the host needs to reach its next launch)."""

    _free = {}   # (device, stream) -> idle mailboxes of this kind (each holds one value at a time)

    def __init__(self, t: torch.Tensor):
        import numpy as np

        assert t.is_cuda and t.element_size() == 8
        self._np = np
        self.shape, self.dtype = tuple(t.shape), t.dtype
        self.nbytes = t.numel() * 8
        self.value = self.mb = None
        if READBACK_MEMCPY or self.nbytes == 0:
            self.value = (t.cpu().numpy() if self.nbytes else
                          np.empty(self.shape, dtype=np.int64 if t.dtype == torch.int64 else np.float64))
            return
        lib = _lib.load()
        t = t.contiguous()
        self.key = (t.device.index, stream_ptr())
        with LAUNCH_LOCK:
            pool = self._free.setdefault(self.key, [])
            mb = None
            for i, cand in enumerate(pool):
                if lib.nvt_mailbox_capacity(cand) >= self.nbytes:
                    mb = pool.pop(i)
                    break
        if mb is None:
            h = C.c_void_p()
            check(lib.nvt_mailbox_create(max(1 << 12, self.nbytes), C.byref(h)), "nvt_mailbox_create")
            mb = h
        self.mb = mb
        seq = C.c_uint64()
        check(lib.nvt_mailbox_post(mb, t.data_ptr(), self.nbytes, stream_ptr(), C.byref(seq)), "nvt_mailbox_post")
        self.seq = seq.value
        self._keep = t

    def __del__(self):
        # never asked for (a fit whose scalars nobody read): the mailbox goes back to the pool --
        # its next post is ordered behind this one on the same stream and carries a later sequence
        mb = getattr(self, "mb", None)
        if mb is not None and getattr(self, "value", None) is None:
            try:
                with LAUNCH_LOCK:
                    self._free.setdefault(self.key, []).append(mb)
            except Exception:   # (interpreter shutdown)
                pass

    def get(self):
        if self.value is None:
            lib = _lib.load()
            check(lib.nvt_mailbox_wait(self.mb, self.seq, READBACK_TIMEOUT_S), "nvt_mailbox_wait")
            np_dt = {torch.int64: self._np.int64, torch.float64: self._np.float64}[self.dtype]
            buf = (C.c_char * self.nbytes).from_address(lib.nvt_mailbox_data(self.mb))
            self.value = self._np.frombuffer(buf, dtype=np_dt).reshape(self.shape).copy()
            with LAUNCH_LOCK:
                self._free.setdefault(self.key, []).append(self.mb)
            self.mb = self._keep = None
        return self.value


def read_back_ptr(ptr: int, nwords: int, device_index: int):
    """uint64[nwords] at device address ``ptr`` -> list of ints, through the mailbox."""
    import numpy as np

    lib = _lib.load()
    nbytes = nwords * 8
    key = (device_index, stream_ptr())
    mb = _mailboxes.get(key)
    if mb is None or lib.nvt_mailbox_capacity(mb) < nbytes:
        if mb is not None:
            lib.nvt_mailbox_destroy(mb)
        h = C.c_void_p()
        check(lib.nvt_mailbox_create(max(1 << 16, 2 * nbytes), C.byref(h)), "nvt_mailbox_create")
        mb = _mailboxes[key] = h
    seq = C.c_uint64()
    check(lib.nvt_mailbox_post(mb, ptr, nbytes, stream_ptr(), C.byref(seq)), "nvt_mailbox_post")
    check(lib.nvt_mailbox_wait(mb, seq.value, READBACK_TIMEOUT_S), "nvt_mailbox_wait")
    buf = (C.c_char * nbytes).from_address(lib.nvt_mailbox_data(mb))
    return np.frombuffer(buf, dtype=np.uint64).astype(np.int64).tolist()


class _timed:
    """Former Python-side event bracket; timing now lives in the library (NVT_PROF scopes)."""

    def __init__(self, name, alg_bytes):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DTYPE_CODE[dt]
    except KeyError:
        raise TypeError(f"unsupported column dtype {dt}") from None


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def aligned(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """Contiguous and 16-byte aligned (views with a storage offset get copied)."""
    if t is None:
        return None
    if not t.is_contiguous():
        t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


MIN_COUNT_CAPACITY = 1 << 16


def next_pow2(x: int) -> int:
    return 1 << max(6, (int(x) - 1).bit_length())


def _key_suffix(keys: torch.Tensor) -> str:
    if keys.dtype == torch.int32:
        return "i32"
    if keys.dtype == torch.int64:
        return "i64"
    raise TypeError(f"categorical keys must be int32/int64 on device, got {keys.dtype}")


# --------------------------------------------------------------------------
# Categorify.fit: count tables
# --------------------------------------------------------------------------
class CountTable:
    """Open-addressing (key -> count) table in HBM for one column (group)."""

    def __init__(self, key_dtype: torch.dtype, capacity: int, device=None):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.key_dtype = key_dtype
        self.key_bytes = 4 if key_dtype == torch.int32 else 8
        self.suffix = "i32" if self.key_bytes == 4 else "i64"
        self.capacity = next_pow2(capacity)
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        nbytes = C.c_uint64()
        check(self.lib.nvt_count_table_bytes(self.key_bytes, self.capacity, C.byref(nbytes)))
        self.table = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
        self.state = torch.zeros(_lib.STATE_WORDS, dtype=torch.int64, device=self.device)
        self.clear()

    def clear(self):
        with _timed("count_clear", 0):
            check(
                self.lib.nvt_count_clear(
                    self.table.data_ptr(), self.key_bytes, self.capacity, self.state.data_ptr(),
                    stream_ptr(),
                ),
                "nvt_count_clear",
            )

    def update(self, keys: torch.Tensor, valid: Optional[torch.Tensor]):
        assert keys.dtype == self.key_dtype
        keys = aligned(keys)
        fn = getattr(self.lib, f"nvt_count_{self.suffix}")
        with _timed(f"count_{self.suffix}", keys.numel() * self.key_bytes):
            check(
                fn(keys.data_ptr(), ptr(valid), keys.numel(), self.table.data_ptr(),
                   self.capacity, self.state.data_ptr(), stream_ptr()),
                f"nvt_count_{self.suffix}",
            )

    def merge(self, keys: torch.Tensor, counts: torch.Tensor):
        assert keys.dtype == self.key_dtype and counts.dtype == torch.int64
        fn = getattr(self.lib, f"nvt_count_merge_{self.suffix}")
        check(
            fn(keys.contiguous().data_ptr(), counts.contiguous().data_ptr(), keys.numel(),
               self.table.data_ptr(), self.capacity, self.state.data_ptr(), stream_ptr()),
            f"nvt_count_merge_{self.suffix}",
        )

    def read_state(self) -> List[int]:
        return self.state.cpu().tolist()  # synchronises the stream

    def compact(self, occupied: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Dense (keys, counts[int64]) in arbitrary order, sentinel-key rows included."""
        st = self.read_state()
        occ = st[_lib.ST_OCCUPIED] if occupied is None else occupied
        out_k = torch.empty(occ + 1, dtype=self.key_dtype, device=self.device)
        out_c = torch.empty(occ + 1, dtype=torch.int64, device=self.device)
        out_n = torch.zeros(1, dtype=torch.int64, device=self.device)
        fn = getattr(self.lib, f"nvt_count_compact_{self.suffix}")
        with _timed("count_compact", 0):
            check(
                fn(self.table.data_ptr(), self.capacity, out_k.data_ptr(), out_c.data_ptr(),
                   out_n.data_ptr(), stream_ptr()),
                f"nvt_count_compact_{self.suffix}",
            )
        n = occ
        if st[_lib.ST_SENTINEL] > 0:
            # rows whose key equals the empty-slot sentinel are counted on the side
            out_k[n] = INT32_MIN if self.key_bytes == 4 else INT64_MIN
            out_c[n] = st[_lib.ST_SENTINEL]
            n += 1
        return out_k[:n], out_c[:n]


def count_into_new_table(
    keys_list: Sequence[torch.Tensor],
    valid_list: Sequence[Optional[torch.Tensor]],
    hint: int,
    max_tries: int = 8,
) -> Tuple[CountTable, List[int]]:
    """Groupby-size of one partition's column(s) into a fresh table sized from
    ``hint`` (expected distinct keys); regrows and recounts on overflow, which
    is safe because the table only holds this partition."""
    total = sum(int(k.numel()) for k in keys_list)
    # never below 2^16 slots: a tiny table puts every block's end-of-kernel flush on the
    # same few memory channels (measured: 36 keys in a 128-slot table cost 340 us of
    # serialised atomics; spread over 512 KiB they cost ~10 us)
    cap = next_pow2(max(MIN_COUNT_CAPACITY, 2 * min(max(hint, 32), max(total, 32))))
    for _ in range(max_tries):
        tab = CountTable(keys_list[0].dtype, cap)
        for k, v in zip(keys_list, valid_list):
            tab.update(k, v)
        st = tab.read_state()
        if not st[_lib.ST_OVERFLOW] and st[_lib.ST_OCCUPIED] * 10 <= tab.capacity * 7:
            return tab, st
        cap = next_pow2(max(16 * cap, 3 * st[_lib.ST_OCCUPIED]))
    raise _lib.NvtHipError("count table kept overflowing; cardinality estimate diverged")


# --------------------------------------------------------------------------
# Categorify.fit, atomic-free: dense (key, count) lists
# --------------------------------------------------------------------------
PATH_S_MAX_DISTINCT = int(os.environ.get("NVT_S_MAX", 11000))          # path 0 (int32 keys, unweighted): 16384-slot LDS tables
PATH_S_MAX_WEIGHTED = 5000          # path 0 for weighted merges / int64 keys: 8192 slots
# Path 7 = path 0 with 2 key classes per row slab (column read twice): 350 us against 420 us on
# path 1 for 12-21 k distinct keys.
# escalation order when a path's LDS tables overflow (C-ABI path ids, include/nvt_hip.h)
PATH_ORDER = [6, 0, 7, 1, 2, 3]
_S_CLASSES = {6: 1, 0: 1, 7: 2}
PATH_S2_FACTOR = 1.95               # path 7: path 0 with 2 key classes per row slab (column read twice)
PATH_TINY_MAX = 64                  # path 6: path 0 with hot keys replicated per lane group
PATH_P1_MAX_DISTINCT = int(os.environ.get("NVT_P1_MAX", 2_400_000))    # path 1: ONE level, 256 buckets x 16384-slot tables (int32)
PATH_P1_MAX_SMALL = 1_100_000       #         ... 8192-slot tables (int64 keys / weighted merges)
PATH_P2_MAX_DISTINCT = 9_000_000    # path 2: 64 x 64 buckets, 4096-slot tables
PATH_P2_MAX_WEIGHTED = 18_000_000   #         weighted: 8192-slot tables
# path 3: 64 x 256 = 16384 buckets, 8192-slot tables that may fill to 6144: at 60 M distinct keys a
# bucket holds 3662 +- 60 of them (Poisson), 40 % headroom -- Criteo-1TB's largest columns
# (C1 / C10 / C20 / C22: 38.5-40.0 M uniques, dask-nvtabular-criteo-benchmark.py:360-366) stay on
# the atomic-free path whether they arrive as one partition or as a merge of partial lists
PATH_P3_MAX_DISTINCT = 60_000_000

# path 9 (range path, include/nvt_hip.h NVT_PATH_RANGE): int32 keys without weights; ONE partition
# pass by key range + per-bucket tables emitted in key order.  Buckets are sized for <= ~5000
# distinct keys (16384-slot tables addressed by a monotone function of the key want a low load).
PATH_RANGE = 9
PATH_PIECES = 0x10000   # include/nvt_hip.h NVT_PATH_PIECES
USE_PIECES = os.environ.get("NVT_RANGE_PIECES", "1") != "0"
RANGE_AUX_WORDS, RANGE_AUX_HIST = 13600 + 256, 8208   # include/nvt_hip.h NVT_RANGE_AUX_*
RANGE_AUX_PW, RANGE_PIECES = 13600, 64   # include/nvt_hip.h NVT_RANGE_AUX_PW, nvt_range.hpp kRpPieces
# distinct keys per bucket the bucket count (256 / 512 / 1024) is chosen for.  Round 6: 10000 (load
# 0.61 of the 16384-slot tables) instead of 5000: fewer bins in the partition pass and range tables
# of half the size -- count 3.90 -> 3.66 ms, ordering 0.87 -> 0.76 ms on cfg2.  A bucket holds at
# most 12288 keys (NVT_OVF_FULL); a column whose keys are not spread evenly enough for that is
# relaunched with all 1024 buckets and REMEMBERED (info["range_bits_floor"] -> Categorify).
RANGE_KEYS_PER_BUCKET = int(os.environ.get("NVT_RANGE_KEYS_PER_BUCKET", "10000"))
PATH_RANGE_MAX_DISTINCT = int(os.environ.get("NVT_RANGE_MAX", 6_500_000))
USE_RANGE = os.environ.get("NVT_RANGE", "1") != "0"

# path 10 (sort path, NVT_PATH_SORT): int32 keys without weights beyond the range path -- radix sort
# of the rows + run lengths, key-sorted output (Criteo-1TB's 38-40 M-unique columns)
PATH_SORT = 10
USE_SORT = os.environ.get("NVT_SORT_PATH", "1") != "0"
USE_FLAT_TABLE = os.environ.get("NVT_FLAT_TABLE", "1") != "0"

_ws_cache = {}
_check_streams = {}   # device index -> side stream of the read-backs that must not wait for work queued later
_RT_BYTES = {}   # range_bits -> nvt_range_table_bytes
_WS_BYTES = {}   # (key_bytes, n, path, weighted) -> nvt_dense_count_ws_bytes


# Columns of one nvt_dense_count_many call that are given different workspaces run on different
# internal streams (include/nvt_hip.h); COUNT_STREAMS workspaces are kept per device.
COUNT_STREAMS = max(1, min(3, int(os.environ.get("NVT_COUNT_STREAMS", "3"))))
# columns that need no hot-key sample start the counting streams (they run under the sample launch)
HEAD_START = os.environ.get("NVT_COUNT_HEAD_START", "1") != "0"
PATH_HOT, HOT_IMAGE_WORDS = 16, 8192   # include/nvt_hip.h NVT_PATH_HOT, NVT_HOT_IMAGE_WORDS
HOT_FILTER = os.environ.get("NVT_HOT_FILTER", "1") != "0"
_PATH_COST = {6: 0.5, 0: 0.7, 7: 1.7, 9: 2.0, 1: 2.5, 2: 3.2, 3: 3.5, 10: 6.0}


def _workspace(nbytes: int, device, slot: int = 0) -> torch.Tensor:
    key = (device.type, device.index, slot)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _ws_cache.pop(key, None)
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _path_for(hint: int, small_tables: bool = False, allow_range: bool = True) -> int:
    s_max = PATH_S_MAX_WEIGHTED if small_tables else PATH_S_MAX_DISTINCT
    if 0 < hint <= PATH_TINY_MAX:
        return 6
    if hint <= s_max:
        return 0
    if hint <= PATH_S2_FACTOR * s_max and (small_tables or not HOT_FILTER):
        return 7  # (with the hot-key filter path 1 is faster there: 241 against 307 us)
    if USE_RANGE and HOT_FILTER and allow_range and not small_tables and hint <= PATH_RANGE_MAX_DISTINCT:
        return PATH_RANGE
    if USE_SORT and not small_tables and hint > PATH_RANGE_MAX_DISTINCT:
        return PATH_SORT
    if USE_SORT and USE_RANGE and HOT_FILTER and not small_tables and not allow_range:
        # a column the range path gave up on (keys that cluster in their range): the sort path
        # keeps its per-partition lists KEY-ORDERED, which is what the partition merge and the
        # one-pass vocabulary ordering live on; the hash paths would hand back unordered lists
        return PATH_SORT
    if hint <= (PATH_P1_MAX_SMALL if small_tables else PATH_P1_MAX_DISTINCT):
        return 1
    if hint <= PATH_P2_MAX_DISTINCT:
        return 2
    if hint <= PATH_P3_MAX_DISTINCT:
        return 3
    return -1  # global-table fallback


class DenseCountJob:
    """One column's groupby-size, launched asynchronously; ``dense_count_many`` reads all
    jobs' state words back with a single device->host copy."""

    def __init__(self, keys, valid, weights=None, hint: int = 0, allow_range: bool = True,
                 pieces=None, min_range_bits: int = 8):
        _lib.require_gpu()
        self.want_table = True    # range path: also dump the count tables as the encode table
        self.range_table = None
        self.allow_range = allow_range  # False: the range path overflowed on this column before
        self.range_failed = False
        self.range_fail_bits = 0
        # range path with a piecewise map (NVT_PATH_PIECES): `pieces` = int32[65] splitters
        # (range_splitters) of a column whose keys are not spread over their range
        self.pieces = pieces if USE_PIECES else None
        # (a column that overflowed its buckets in an earlier fit starts with the bucket count
        # that held it: the caller remembers info["range_bits_floor"])
        self.min_range_bits = max(8, min(10, int(min_range_bits)))
        self.lib = _lib.load()
        self.keys = aligned(keys)
        self.valid = valid
        self.weights = weights.contiguous() if weights is not None else None
        self.n = int(self.keys.numel())
        self.dev = self.keys.device
        self.suffix = _key_suffix(self.keys)
        self.kb = 4 if self.suffix == "i32" else 8
        self.hint = hint
        self.path = _path_for(hint, small_tables=(weights is not None or self.kb == 8),
                              allow_range=allow_range)
        self.cap_guess = max(1 << 16, 2 * max(hint, 1))
        self.state = None  # device uint64[STATE_WORDS] view, assigned by dense_count_many
        self.result = None
        self.hot = None    # None: HOT_FILTER decides; True / False: forced (tests, probes)

    def range_bits(self) -> int:
        """log2 of the bucket count of the range path: 256 .. 1024 buckets."""
        want = max(1, -(-max(self.hint, 1) // RANGE_KEYS_PER_BUCKET))
        return max(self.min_range_bits, min(10, max(8, (want - 1).bit_length())))

    def _launch_path_of(self, path: int) -> int:
        if path == PATH_RANGE:
            return PATH_RANGE | (self.range_bits() << 8) | (PATH_PIECES if self.pieces is not None else 0)
        eligible = path in (1, 2, 3) and self.kb == 4 and self.weights is None
        hot = HOT_FILTER if self.hot is None else self.hot
        return path | PATH_HOT if (eligible and hot) else path

    def _launch_path(self) -> int:
        """The `path` argument of the C call: partitioned paths of int32 keys without weights
        get the hot-key filter in front (include/nvt_hip.h, NVT_PATH_HOT)."""
        return self._launch_path_of(self.path)

    def prepare(self, desc: "_lib.CountCol") -> int:
        """Allocate this attempt's output list and fill one nvt_count_col descriptor; returns
        the workspace bytes the path needs (the caller shares ONE workspace per call)."""
        n, path = self.n, self.path
        out_cap = min(self.cap_guess, n) + 1
        if path in _S_CLASSES:
            # stage 2 has 256 workgroups per key class, each emitting <= 384 keys
            out_cap = min(out_cap, _S_CLASSES[path] * 256 * 384 + 1)
        self.out_k = torch.empty(out_cap + 1, dtype=self.keys.dtype, device=self.dev)
        self.out_c = torch.empty(out_cap + 1, dtype=torch.int64, device=self.dev)
        lpath = self._launch_path()
        key = (self.kb, n, lpath, self.weights is not None)
        nbytes = _WS_BYTES.get(key)
        if nbytes is None:
            out = C.c_uint64()
            check(self.lib.nvt_dense_count_ws_bytes(self.kb, n, lpath,
                                                    0 if self.weights is None else 1, C.byref(out)))
            nbytes = _WS_BYTES[key] = out.value
        desc.keys = self.keys.data_ptr()
        desc.valid = ptr(self.valid)
        desc.weights = ptr(self.weights)
        desc.n = n
        desc.key_bytes = self.kb
        desc.path = lpath
        desc.out_keys = self.out_k.data_ptr()
        desc.out_counts = self.out_c.data_ptr()
        desc.out_capacity = out_cap
        desc.state = self.state.data_ptr()
        desc.hot_image = None
        desc.range_table = None
        if path == PATH_SORT:
            # uint32[256] block that receives the histogram of min(count, 255)
            self.hot_image = torch.empty(256, dtype=torch.int32, device=self.dev)
            desc.hot_image = self.hot_image.data_ptr()
        elif path == PATH_RANGE:
            # aux block: hot image + range parameters (sample kernel) + class histogram
            self.hot_image = torch.empty(RANGE_AUX_WORDS, dtype=torch.int32, device=self.dev)
            if self.pieces is not None:
                self.hot_image[RANGE_AUX_PW:RANGE_AUX_PW + RANGE_PIECES + 1] = self.pieces
            desc.hot_image = self.hot_image.data_ptr()
            if self.want_table:
                # the per-bucket count tables, dumped: the column's encode table (range table)
                bits = self.range_bits()
                rt_bytes = _RT_BYTES.get(bits)
                if rt_bytes is None:
                    rt_out = C.c_uint64()
                    check(self.lib.nvt_range_table_bytes(bits, C.byref(rt_out)))
                    rt_bytes = _RT_BYTES[bits] = rt_out.value
                self.range_table = torch.empty(rt_bytes, dtype=torch.uint8, device=self.dev)
                self.table_bits = bits
                desc.range_table = self.range_table.data_ptr()
        elif lpath & PATH_HOT:
            # the column's own hot-key table image: sampled for all columns by one launch
            self.hot_image = torch.empty(HOT_IMAGE_WORDS, dtype=torch.int32, device=self.dev)
            desc.hot_image = self.hot_image.data_ptr()
        return nbytes

    def resolve(self, st) -> bool:
        """Inspect the state words read back for this job; False = relaunch needed."""
        ovf = st[_lib.ST_OVERFLOW]
        if ovf & 1:
            order = PATH_ORDER
            if self.path == PATH_SORT:  # (no LDS tables to overflow)
                raise _lib.NvtHipError("dense count: the sort path reported a table overflow")
            if self.path == PATH_RANGE and self.range_bits() < 10:
                # more distinct keys than the hint promised (cold start): all 1024 buckets first
                self.min_range_bits = 10
                self.cap_guess = max(self.cap_guess, 1 << 22)
                return False
            # (the range path assumes keys spread over their range; a hash path does not)
            nxt = order.index(1) if self.path == PATH_RANGE else order.index(self.path) + 1
            if self.path == PATH_RANGE and not self.range_failed:
                self.range_fail_bits = int(ovf)  # NVT_OVF_*: region / probe / full (diagnostic)
            self.range_failed = self.range_failed or self.path == PATH_RANGE
            if (self.path == 0 and USE_RANGE and HOT_FILTER and self.allow_range and self.kb == 4
                    and self.weights is None and not self.range_failed):
                # LDS tables too small: the range path comes next for int32 keys
                self.path = PATH_RANGE
                self.cap_guess = max(self.cap_guess, 1 << 20)
                return False
            if nxt >= len(order):
                self._fallback()
                return True
            self.path = order[nxt]
            if self.path == 7 and self._launch_path_of(1) & PATH_HOT:
                self.path = 1  # the filtered path 1 beats two key classes (see _path_for)
            if self.path == 1 and self.range_failed and USE_SORT and self.kb == 4 and self.weights is None:
                # the range path gave up at 1024 buckets: more keys than it holds, or keys that
                # are not spread over their range -- the sort path copes with both in one go
                # (the next fit picks its path from the distinct count found here)
                self.path = PATH_SORT
            elif self.path == 1 and self.hint > PATH_P1_MAX_DISTINCT:
                self.path = 2 if self.hint <= PATH_P2_MAX_DISTINCT else 3
            self.cap_guess = max(self.cap_guess, _PATH_MAX[self.path])
            return False
        if ovf & 2:
            if self.cap_guess > self.n:
                raise _lib.NvtHipError("dense count: output list overflowed at full capacity")
            need = st[_lib.ST_NEED] if self.path in (PATH_RANGE, PATH_SORT) else 0
            # (the range path reports the exact size; the hash paths only that it was too small)
            self.cap_guess = need + 64 if need > self.cap_guess else max(4 * self.cap_guess, 1 << 20)
            return False
        m = st[_lib.ST_OCCUPIED]
        max_count = st[_lib.ST_MAXCOUNT]
        if self.path == PATH_SORT:
            self.result = (self.out_k[:m], self.out_c[:m], st[_lib.ST_NULLS],
                           dict(path=self.path, distinct=m, max_count=max_count,
                                rows=st[_lib.ST_ROWS], sorted_by_key=True, cls_hist=self.hot_image,
                                n_big=st[_lib.ST_BIG], range_failed=self.range_failed,
                                range_fail_bits=self.range_fail_bits,
                                range_pieces=self.pieces is not None))
            return True
        if self.path == PATH_RANGE:
            # key-ordered list (the sentinel key, smallest int32, already leads it) + what the
            # one-pass vocabulary ordering needs: histogram of min(count, 255), entries >= 255
            self.result = (self.out_k[:m], self.out_c[:m], st[_lib.ST_NULLS],
                           dict(path=self.path, distinct=m, max_count=max_count,
                                rows=st[_lib.ST_ROWS], sorted_by_key=True,
                                cls_hist=self.hot_image[RANGE_AUX_HIST:RANGE_AUX_HIST + 256],
                                n_big=st[_lib.ST_BIG], range_aux=self.hot_image,
                                range_table=self.range_table,
                                range_bits=self.table_bits if self.range_table is not None else 0,
                                range_pieces=self.pieces is not None,
                                range_fail_bits=self.range_fail_bits,
                                range_bits_floor=self.min_range_bits))
            return True
        if st[_lib.ST_SENTINEL] > 0:
            self.out_k[m] = INT32_MIN if self.kb == 4 else INT64_MIN
            self.out_c[m] = st[_lib.ST_SENTINEL]
            max_count = max(max_count, st[_lib.ST_SENTINEL])
            m += 1
        self.result = (self.out_k[:m], self.out_c[:m], st[_lib.ST_NULLS],
                       dict(path=self.path, distinct=m, max_count=max_count, rows=st[_lib.ST_ROWS],
                            range_failed=self.range_failed, range_fail_bits=self.range_fail_bits))
        return True

    def _fallback(self):
        """Last resort: one global open-addressing table (device atomics)."""
        if self.weights is None:
            tab, st = count_into_new_table([self.keys], [self.valid], max(self.hint, 1 << 20))
        else:
            tab = CountTable(self.keys.dtype, 2 * self.n)
            tab.merge(self.keys, self.weights)
            st = tab.read_state()
        k, c = tab.compact()
        mx = int(c.max().item()) if c.numel() else 0
        self.result = (k, c, st[_lib.ST_NULLS],
                       dict(path=-1, distinct=int(k.numel()), max_count=mx, rows=self.n))


SAMPLE_ROWS = 1 << 18               # cold start: distinct keys of this many leading rows ...
SAMPLE_MIN_ROWS = 8 * SAMPLE_ROWS   # ... when the column is at least this long
# distinct keys each path is sized for (output-capacity guess when a path is entered by escalation)
_PATH_MAX = {6: 1024, 0: 98304, 7: 196608, 9: PATH_RANGE_MAX_DISTINCT, 10: 1 << 30, 1: PATH_P1_MAX_DISTINCT,
             2: PATH_P2_MAX_DISTINCT, 3: PATH_P3_MAX_DISTINCT}


def _estimate_distinct(d: int, m: int, n: int) -> int:
    """Distinct keys expected in n rows given d distinct among the first m (uniform-draw
    model d = D (1 - exp(-m / D)), tripled for the heavy tails of real categorical data).
    Only steers the first path choice: an underestimate costs one cheap failed attempt."""
    if m <= 0 or d <= 0:
        return 1
    r = d / m
    if r < 0.02:
        D = float(d)
    elif r > 0.999:
        D = float(n)
    else:
        lo, hi = 1e-6, 60.0  # x = m / D, (1 - exp(-x)) / x decreases from 1 to 0
        for _ in range(60):
            mid = 0.5 * (lo + hi)
            if (1.0 - math.exp(-mid)) / mid > r:
                lo = mid
            else:
                hi = mid
        D = m / (0.5 * (lo + hi))
    return int(min(n, 3.0 * D + 64))


STATS = {"count_relaunches": 0, "presampled_columns": 0}  # diagnostics (bench.py cold step)
PRESAMPLE_SKETCH = os.environ.get("NVT_PRESAMPLE_EXACT", "0") != "1"


def _presample(jobs):
    """Jobs with no cardinality hint (first partition of a fit) start on a path chosen from
    the distinct count of a 256 K-row prefix instead of climbing PATH_ORDER from the bottom:
    the cold fit of the 45 M-row Criteo frame drops from 74 ms to under 30 ms."""
    cand = [j for j in jobs if j.result is None and j.hint <= 0 and j.weights is None
            and j.n >= SAMPLE_MIN_ROWS]
    if not cand:
        return
    if PRESAMPLE_SKETCH:
        # ONE launch: a HyperLogLog sketch of every column's prefix (nvt_prefix_distinct); the exact
        # count below went through ~11 launches per column for numbers that are tripled anyway
        descs = (_lib.PrefixCol * len(cand))()
        for d, j in zip(descs, cand):
            d.keys, d.valid = j.keys.data_ptr(), ptr(j.valid)
            d.n, d.key_bytes = min(SAMPLE_ROWS, j.n), j.kb
        out = torch.empty((len(cand), 2), dtype=torch.int64, device=cand[0].dev)
        check(cand[0].lib.nvt_prefix_distinct(descs, len(cand), out.data_ptr(), stream_ptr()),
              "nvt_prefix_distinct")
        res = [(None, None, 0, dict(distinct=int(d), rows=int(r))) for d, r in read_back(out).tolist()]
    else:
        samples = []
        for j in cand:
            valid = None if j.valid is None else j.valid[: SAMPLE_ROWS // 8]
            sj = DenseCountJob(j.keys[:SAMPLE_ROWS], valid, None, hint=SAMPLE_ROWS)
            sj.path, sj.cap_guess = 1, SAMPLE_ROWS
            samples.append(sj)
        res = _run_jobs(samples)
    for j, (_, _, nulls, info) in zip(cand, res):
        est = _estimate_distinct(info["distinct"], info["rows"] - nulls, j.n)
        j.hint = est
        stat_add("presampled_columns")
        # an estimate, not a hint: the range path starts with all its buckets (a column with more
        # keys than estimated would overflow 256 / 512 buckets and be counted twice) -- unless the
        # prefix shows every key >= 10 times on average: the estimate (3 x the uniform model) then
        # sizes the buckets.  (Heavy tails defeat any extrapolation from 0.6 % of the rows: a
        # Criteo column with 6.2 M keys shows 98 k in the prefix and is estimated at 322 k, Chao1
        # says 550 k; such columns show most prefix keys once or twice and keep 1024 buckets.)
        seen_rows = max(int(info["rows"]) - int(nulls), 1)
        j.min_range_bits = max(j.min_range_bits, 10 if int(info["distinct"]) * 10 >= seen_rows else 8)
        j.path = _path_for(est, small_tables=(j.kb == 8), allow_range=j.allow_range)
        # ... and a roomy output list (a sixth of the rows): a relaunch costs a full recount
        j.cap_guess = max(1 << 16, 2 * est, j.n // 6 if j.path not in _S_CLASSES else 0)


class CountBatch:
    """Every column's groupby-size of one partition, enqueued by ONE C call
    (nvt_dense_count_many); ``results()`` performs the single device->host read of all the
    state words and relaunches the (rare, once hints are learned) columns whose LDS tables or
    output lists overflowed.  Launch and read-back are separate so that the caller can queue
    other work (Normalize's moments, the next partition's copies) before synchronising."""

    def __init__(self, jobs):
        self.jobs = list(jobs)
        self._results = None
        _presample(self.jobs)
        for j in self.jobs:
            if j.n == 0:
                j.result = (torch.empty(0, dtype=j.keys.dtype, device=j.dev),
                            torch.empty(0, dtype=torch.int64, device=j.dev), 0,
                            dict(path=0, distinct=0, max_count=0, rows=0))
        for j in self.jobs:
            if j.result is None and j.path < 0:
                j._fallback()
        self.pending = [j for j in self.jobs if j.result is None]
        self.states = None
        self._launch()

    def _launch(self):
        pending = self.pending
        if not pending:
            return
        if COUNT_STREAMS > 1 and len(pending) >= 2 and HEAD_START:
            # The C call launches the columns in the order of the descriptors, each on the stream of
            # its workspace, and a stream waits for the hot-key samples in front of its first
            # FILTERED column.  In column order every stream began with a range-path column: 243
            # CUs idled for the ~110 us of the sample launch.  COUNT_STREAMS columns that need no
            # sample (LDS-resident: paths 0 / 6 / 7) go first -- they land on different streams
            # (longest-processing-time assignment below: equal loads at that point) and run
            # under the sample.
            free = [j for j in pending if j.path in _S_CLASSES][:COUNT_STREAMS]
            if free and len(free) < len(pending):
                ids = {id(j) for j in free}
                pending = self.pending = free + [j for j in pending if id(j) not in ids]
        dev = pending[0].dev
        self.states = torch.empty(len(pending), _lib.STATE_WORDS, dtype=torch.int64, device=dev)
        descs = (_lib.CountCol * len(pending))()
        need = 0
        for i, j in enumerate(pending):
            j.state = self.states[i]
            need = max(need, j.prepare(descs[i]))
        if COUNT_STREAMS == 1 or len(pending) < 2:
            wp = _workspace(need, dev).data_ptr()  # shared: the columns are ordered on one stream
            for d in descs:
                d.ws = wp
        else:
            # longest-processing-time assignment of the columns to COUNT_STREAMS workspaces
            load = [0.0] * COUNT_STREAMS
            wps = [_workspace(need, dev, k).data_ptr() for k in range(COUNT_STREAMS)]
            order = sorted(range(len(pending)),
                           key=lambda i: -_PATH_COST.get(pending[i].path, 1.0) * pending[i].n)
            head = 0
            if HEAD_START:   # (the head-start columns: one per stream, whatever their cost)
                while head < min(COUNT_STREAMS, len(pending)) and pending[head].path in _S_CLASSES \
                        and pending[-1].path not in _S_CLASSES:
                    head += 1
                if head < COUNT_STREAMS:
                    head = 0
            for i in range(head):
                load[i] += _PATH_COST.get(pending[i].path, 1.0) * pending[i].n
                descs[i].ws = wps[i]
            for i in order:
                if i < head:
                    continue
                k = load.index(min(load))
                load[k] += _PATH_COST.get(pending[i].path, 1.0) * pending[i].n
                descs[i].ws = wps[k]
        check(_lib.load().nvt_dense_count_many(descs, len(pending), stream_ptr()),
              "nvt_dense_count_many")
        # what results() waits for: THIS call's kernels, not whatever is queued behind them (the
        # next partition's counting is launched before this one is read back)
        self._done = torch.cuda.Event()
        self._done.record()

    def _read_states(self):
        dev = self.states.device
        side = _check_streams.get(dev.index)
        if side is None:
            side = _check_streams[dev.index] = torch.cuda.Stream(device=dev)
        side.wait_event(self._done)
        with torch.cuda.stream(side):
            return read_back(self.states).tolist()

    def results(self):
        if self._results is None:
            while self.pending:
                host = self._read_states()  # the single synchronisation point
                self.pending = [j for i, j in enumerate(self.pending) if not j.resolve(host[i])]
                stat_add("count_relaunches", len(self.pending))
                self._launch()
            self._results = [j.result for j in self.jobs]
        return self._results


def dense_count_many(jobs):
    """Launch every job (one C call), then ONE readback for all their state words."""
    return CountBatch(jobs).results()


def _run_jobs(jobs):
    batch = CountBatch.__new__(CountBatch)
    batch.jobs = list(jobs)
    batch._results = None
    for j in batch.jobs:
        if j.n == 0:
            j.result = (torch.empty(0, dtype=j.keys.dtype, device=j.dev),
                        torch.empty(0, dtype=torch.int64, device=j.dev), 0,
                        dict(path=0, distinct=0, max_count=0, rows=0))
    for j in batch.jobs:
        if j.result is None and j.path < 0:
            j._fallback()
    batch.pending = [j for j in batch.jobs if j.result is None]
    batch.states = None
    batch._launch()
    return batch.results()


def dense_count(
    keys: torch.Tensor,
    valid: Optional[torch.Tensor],
    weights: Optional[torch.Tensor] = None,
    hint: int = 0,
):
    """Groupby-size (or weighted sum) of a key column -> (keys, counts[int64], nulls, info).

    ``hint`` = expected number of distinct keys (0 = unknown).  Picks the LDS /
    partitioned path from it and escalates when a kernel reports that its LDS
    tables filled up; the last resort is the global-table kernel of nvt_count_*.
    info = dict(path, distinct, max_count, rows) so callers can remember the hint."""
    return dense_count_many([DenseCountJob(keys, valid, weights, hint)])[0]


def merge_dense(lists, hint: int = 0):
    """Tree-merge step (_mid_level_groupby): sum the counts of equal keys across
    several dense (keys, counts) lists."""
    lists = [(t[0], t[1]) + tuple(t[2:]) for t in lists if t[0].numel()]
    if not lists:
        return None
    if len(lists) == 1:
        return lists[0] if len(lists[0]) == 3 else (lists[0][0], lists[0][1], 0)
    lists = [(t[0], t[1]) for t in lists]
    dt = torch.int64 if any(k.dtype == torch.int64 for k, _ in lists) else torch.int32
    keys = torch.cat([k.to(dt) for k, _ in lists])
    counts = torch.cat([c for _, c in lists])
    k, c, _, info = dense_count(keys, None, counts,
                                hint=hint or max(int(x[0].numel()) for x in lists))
    return k, c, info["max_count"]


def merge_dense_many(groups, hints=None):
    """merge_dense for several groups with ONE nvt_dense_count_many call (and one read-back):
    groups[j] = the (keys, counts[, max_count]) lists of group j -> [(keys, counts, max_count)]."""
    jobs, slots, out = [], [], [None] * len(groups)
    for j, lists in enumerate(groups):
        lists = [(t[0], t[1]) for t in lists if t[0].numel()]
        if not lists:
            continue
        if len(lists) == 1:
            out[j] = (lists[0][0], lists[0][1], int(groups[j][0][2]) if len(groups[j][0]) > 2 else 0)
            continue
        dt = torch.int64 if any(k.dtype == torch.int64 for k, _ in lists) else torch.int32
        keys = torch.cat([k.to(dt) for k, _ in lists])
        counts = torch.cat([c for _, c in lists])
        hint = (hints[j] if hints else 0) or max(int(x[0].numel()) for x in lists)
        jobs.append(DenseCountJob(keys, None, counts, hint))
        slots.append(j)
    if jobs:
        for j, (k, c, _, info) in zip(slots, dense_count_many(jobs)):
            out[j] = (k, c, info["max_count"])
    return out


MERGE_SORTED = os.environ.get("NVT_MERGE_SORTED", "1") != "0"
MERGE_SORTED_MAX_TOTAL = (1 << 31) - 1   # include/nvt_hip.h nvt_merge_sorted_many: na + nb < 2^31


def merge_sorted_pairs(pairs, want_src: bool = False):
    """nvt_merge_sorted_many: pairs = [((keys_a, counts_a), (keys_b, counts_b))] of KEY-SORTED,
    duplicate-free int32 lists (counts int64 or None) -> [(keys, counts)] -- the union in key
    order, counts of equal keys summed; with want_src [(keys, counts, src_a, src_b)] where
    src_* = the position in A / B an output entry came from (-1: none).  ONE merge-path launch +
    ONE tile launch for all pairs, ONE read-back (the merged lengths)."""
    if not pairs:
        return []
    _lib.require_gpu()
    lib = _lib.load()
    dev = pairs[0][0][0].device
    descs = (_lib.MergeCol * len(pairs))()
    out_n = torch.empty(len(pairs), dtype=torch.int64, device=dev)
    outs, keep = [], []
    # ONE output buffer per array for the whole call (a multi-GPU owner merges ~100 pairs per
    # level: four allocations per pair were most of the call), every pair on a 16-byte boundary
    sizes, offs, tot = [], [], 0
    for (ka, _), (kb, _) in pairs:
        n_ab = int(ka.numel()) + int(kb.numel())
        if n_ab > MERGE_SORTED_MAX_TOTAL:
            raise _lib.NvtHipError("merge_sorted_pairs: more than 2^31 - 1 entries in one merge")
        sizes.append(n_ab)
        offs.append(tot)
        tot += (max(n_ab, 1) + 3) // 4 * 4   # (at least one element: an empty pair must still look
                                            # like "with src maps" to the library)
    with_counts = any(ca is not None or cb is not None for (_, ca), (_, cb) in pairs)
    ok_all = torch.empty(tot, dtype=torch.int32, device=dev)
    oc_all = torch.empty(tot, dtype=torch.int64, device=dev) if with_counts else None
    sa_all = torch.empty(tot, dtype=torch.int32, device=dev) if want_src else None
    sb_all = torch.empty(tot, dtype=torch.int32, device=dev) if want_src else None
    pk, pc = ok_all.data_ptr(), oc_all.data_ptr() if oc_all is not None else 0
    psa, psb = (sa_all.data_ptr(), sb_all.data_ptr()) if want_src else (0, 0)
    pn = out_n.data_ptr()
    for i, (d, ((ka, ca), (kb, cb))) in enumerate(zip(descs, pairs)):
        assert ka.dtype == torch.int32 and kb.dtype == torch.int32
        ka, kb = ka.contiguous(), kb.contiguous()
        ca = ca.contiguous() if ca is not None else None
        cb = cb.contiguous() if cb is not None else None
        na, nb = int(ka.numel()), int(kb.numel())
        has_c = ca is not None or cb is not None
        o = offs[i]
        d.a_keys, d.a_counts, d.na = ptr(ka) if na else None, ptr(ca) if na else None, na
        d.b_keys, d.b_counts, d.nb = ptr(kb) if nb else None, ptr(cb) if nb else None, nb
        d.out_keys, d.out_counts = pk + 4 * o, (pc + 8 * o) if has_c else None
        d.src_a, d.src_b = (psa + 4 * o, psb + 4 * o) if want_src else (None, None)
        d.out_n = pn + 8 * i
        outs.append((o, has_c))
        keep.append((ka, kb, ca, cb))
    need = C.c_uint64()
    check(lib.nvt_merge_sorted_ws_bytes(descs, len(pairs), C.byref(need)), "nvt_merge_sorted_ws_bytes")
    ws = torch.empty(need.value + 16, dtype=torch.uint8, device=dev)
    check(lib.nvt_merge_sorted_many(descs, len(pairs), ws.data_ptr(), need.value, stream_ptr()),
          "nvt_merge_sorted_many")
    ns = read_back(out_n).tolist()
    res = []
    for (o, has_c), n in zip(outs, ns):
        n = int(n)
        if want_src:
            res.append((ok_all[o:o + n], oc_all[o:o + n] if has_c else None, sa_all[o:o + n], sb_all[o:o + n]))
        else:
            res.append((ok_all[o:o + n], oc_all[o:o + n] if has_c else None))
    return res


def merge_payload(src_a, src_b, a, b, op: str = "add", width: int = 1):
    """nvt_merge_payload: rows of ``width`` int64 / float64 values of two merged lists combined
    through the src maps of merge_sorted_pairs (op: add / min / max; a missing side contributes
    nothing).  a, b: [entries * width] contiguous."""
    n = int(src_a.numel())
    a, b = a.contiguous(), b.contiguous()
    assert a.dtype == b.dtype and a.dtype in (torch.int64, torch.float64)
    out = torch.empty(n * width, dtype=a.dtype, device=src_a.device)
    check(_lib.load().nvt_merge_payload(
        ptr(src_a), ptr(src_b), n, int(width), dtype_code(a.dtype), {"add": 0, "min": 1, "max": 2}[op],
        ptr(a) if a.numel() else None, ptr(b) if b.numel() else None, out.data_ptr(), stream_ptr()),
        "nvt_merge_payload")
    return out


def merge_sorted_tree(col_lists):
    """Tree merge (_mid_level_groupby, categorify.py:1054-1070, over the tree of :1423-1478) of
    KEY-SORTED (keys int32, counts int64) lists: col_lists[j] = the lists of column j; returns
    one (keys, counts) per column.  Every level merges disjoint pairs of ALL columns in one
    call (nvt_merge_sorted_many) and costs one read-back; a list much larger than the others
    (the table accumulated so far) is held back for the last level, so it is streamed once."""
    cur = [[(k, c) for k, c in lists if int(k.numel())] for lists in col_lists]
    while any(len(l) > 1 for l in cur):
        pairs, plan = [], []
        for j, lists in enumerate(cur):
            if len(lists) < 2:
                plan.append((j, [], lists))
                continue
            order = sorted(lists, key=lambda t: int(t[0].numel()))
            held = []
            if len(order) > 2 and int(order[-1][0].numel()) > 2 * int(order[-2][0].numel()):
                held = [order.pop()]
            mine = []
            while len(order) >= 2:
                a, b = order.pop(0), order.pop(0)
                mine.append(len(pairs))
                pairs.append((a, b))
            plan.append((j, mine, order + held))
        merged = merge_sorted_pairs(pairs)
        cur = [[merged[i] for i in mine] + rest for j, mine, rest in plan]
    devs = [t[0].device for lists in col_lists for t in lists]
    dev = devs[0] if devs else torch.device("cuda", torch.cuda.current_device())
    out = []
    for lists in cur:
        out.append(lists[0] if lists else (torch.empty(0, dtype=torch.int32, device=dev),
                                           torch.empty(0, dtype=torch.int64, device=dev)))
    return out


def range_splitters(keys: torch.Tensor, counts: torch.Tensor, rows: int = 0):
    """int32[65] splitters of the piecewise range map (NVT_PATH_PIECES) from an EXACT key-ordered
    (key, count) list of the column (what the sort path returns after the linear map
    overflowed), or None when the list is too short to need them.

    The range path wants buckets with at most ~2x the average of (a) the rows that go through
    the partition bins and (b) the distinct keys of a bucket table.  Every key gets the weight
    1/3 x its share of the bin rows (the ~8192 most frequent keys, which the hot image mostly
    absorbs, count with a third of their rows) + 2/3 x its share of the distinct keys (1 / entries); splitter p
    is the key at which the running weight reaches p / 64.  Values are the order-preserving
    uint32 images of the keys (key ^ 2^31), stored in an int32 tensor.  One read-back."""
    n = int(keys.numel())
    if n < 4096 or keys.dtype != torch.int32:
        return None
    k64 = keys.to(torch.int64)
    if int(k64[0].item()) == INT32_MIN:   # the sentinel key leads the list and is not in any table
        k64, counts, n = k64[1:], counts[1:], n - 1
    c = counts.to(torch.float64)
    rows = rows or float(c.sum().item())
    # rows that go through the bins = rows of the keys that are NOT in the hot image.  The image
    # (hot_sample_kernel) takes the keys the 64 K-row sample shows twice, the most frequent ones
    # first; a key with lam expected sample hits is in with about P(Poisson(lam) >= 2), less the
    # bucket conflicts of the late-comers (2-slot buckets: ~15 %); lam >= 8: practically always
    lam = c * (65536.0 / max(float(rows), 1.0))
    p_in = torch.where(lam >= 8.0, torch.full_like(c, 0.98),
                       0.85 * (1.0 - torch.exp(-lam) * (1.0 + lam)))
    w_rows = c * (1.0 - p_in)
    tot = w_rows.sum()
    w = (2.0 / 3.0) / n + torch.where(tot > 0, w_rows / torch.clamp(tot, min=1.0) / 3.0, torch.zeros_like(c) + (1.0 / 3.0) / n)
    cum = torch.cumsum(w, 0)
    cum = cum / cum[-1]
    targets = torch.arange(1, RANGE_PIECES, dtype=torch.float64, device=keys.device) / RANGE_PIECES
    idx = torch.searchsorted(cum, targets).clamp_(max=n - 1)
    u = (torch.cat([k64[:1], k64[idx], k64[-1:] + 1]) + (1 << 31))
    host = read_back(u).tolist()
    for p in range(1, RANGE_PIECES + 1):      # strictly increasing (degenerate pieces: one key wide)
        if host[p] <= host[p - 1]:
            host[p] = host[p - 1] + 1
    if host[-1] > 0xFFFFFFFF:
        return None
    arr = torch.tensor(host, dtype=torch.int64, device=keys.device)
    return torch.where(arr >= (1 << 31), arr - (1 << 32), arr).to(torch.int32)   # the u32 bit pattern


def class_hist(counts: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """int32[256] histogram of min(count, 255) on the device (nvt_class_hist): the input of the
    one-pass ordering of a key-sorted (key, count) list.  out: a contiguous int32[256] to fill."""
    hist = out if out is not None else torch.empty(256, dtype=torch.int32, device=counts.device)
    counts = counts.contiguous()
    check(_lib.load().nvt_class_hist(ptr(counts) if counts.numel() else None, counts.numel(),
                                     hist.data_ptr(), stream_ptr()), "nvt_class_hist")
    return hist


def label_shard(keys: torch.Tensor, counts: torch.Tensor, class_base_diff: torch.Tensor, n_big: int,
                label_of: torch.Tensor):
    """nvt_vocab_label_shard: positions in the vocabulary order of the UNION for this rank's shard
    of a key-sorted (key, count) list (entries with count < 255; the others -1), written into
    ``label_of`` (int32[n]); returns (big_keys, big_counts, big_src) of the n_big entries with
    count >= 255, compacted in key order (big_src: their positions in the shard)."""
    n = int(keys.numel())
    dev = keys.device
    bk = torch.empty(max(n_big, 1), dtype=torch.int32, device=dev)
    bc = torch.empty(max(n_big, 1), dtype=torch.int64, device=dev)
    bs = torch.empty(max(n_big, 1), dtype=torch.int32, device=dev)
    if n == 0:
        return bk[:0], bc[:0], bs[:0]
    lib = _lib.load()
    out = C.c_uint64()
    check(lib.nvt_vocab_order_tmp_bytes(n, 0, C.byref(out)), "nvt_vocab_order_tmp_bytes")
    tmp = torch.empty(out.value + 16, dtype=torch.uint8, device=dev)
    check(lib.nvt_vocab_label_shard(keys.contiguous().data_ptr(), counts.contiguous().data_ptr(), n,
                                    class_base_diff.data_ptr(), tmp.data_ptr(), label_of.data_ptr(),
                                    bk.data_ptr(), bc.data_ptr(), bs.data_ptr(), stream_ptr()),
          "nvt_vocab_label_shard")
    return bk[:n_big], bc[:n_big], bs[:n_big]


def sort_by_key(keys: torch.Tensor, counts: torch.Tensor):
    """(keys, counts) ordered by key ascending (radix sort of nvt_order_rows)."""
    n = int(keys.numel())
    if n <= 1:
        return keys, counts
    perm = order_rows(n, keys.device, sort_keys=[(keys, None, True)]) & 0xFFFFFFFF
    return keys[perm].contiguous(), counts[perm].contiguous()


def vocab_sort(keys: torch.Tensor, counts: torch.Tensor, max_count: int = 0):
    """In place: (count desc, key asc) -- categorify.py:1300,1316 with the stable tie rule.
    max_count (upper bound on counts, 0 = unknown) avoids a stream synchronisation."""
    lib = _lib.load()
    n = keys.numel()
    if n <= 1:
        return
    suffix = _key_suffix(keys)
    nbytes = C.c_uint64()
    check(lib.nvt_vocab_sort_tmp_bytes(4 if suffix == "i32" else 8, n, C.byref(nbytes)))
    tmp = torch.empty(nbytes.value + 16, dtype=torch.uint8, device=keys.device)
    assert keys.is_contiguous() and counts.is_contiguous()
    with _timed("vocab_sort", 0):
        check(
            getattr(lib, f"nvt_vocab_sort_{suffix}")(
                keys.data_ptr(), counts.data_ptr(), n, int(max_count), tmp.data_ptr(), stream_ptr()
            ),
            "nvt_vocab_sort",
        )


# --------------------------------------------------------------------------
# Categorify.transform: encode tables
# --------------------------------------------------------------------------
ENCODE_RESIDENT_I32, ENCODE_RESIDENT_I64 = 8192, 6144  # include/nvt_hip.h
_ENC_BYTES = {}   # (key_bytes, capacity) -> nvt_encode_table_bytes
ASYNC_FINALIZE = os.environ.get("NVT_ASYNC_FINALIZE", "1") != "0"
_SORT_BYTES = {}  # (key_bytes, n) -> nvt_vocab_sort_tmp_bytes


ENCODE_HEAD_IMAGE = os.environ.get("NVT_ENCODE_HEAD_IMAGE", "1") != "0"


class EncodeTable:
    """key -> label probe table built from an ordered vocabulary."""

    FLAT_AUX_WORDS, FLAT_AUX_MAXDISP = 8192 + 16, 8192 + 8   # include/nvt_hip.h NVT_FLAT_AUX_*
    FLAT_MAX_DISPLACEMENT = 4096   # longer probe runs than this: the keys cluster, use a hashed table

    def __init__(self, vocab_keys: torch.Tensor, first_label: int, unique: bool = False,
                 defer_build: bool = False, range_table=None, flat: bool = False):
        """range_table = (table, aux, bits): the table was dumped by the range path of the
        counting stage and is addressed by the monotone map stored in ``aux``; it is completed
        (positions -> labels) by nvt_vocab_finalize_many -- no table is allocated or built here."""
        _lib.require_gpu()
        self.lib = _lib.load()
        self.suffix = _key_suffix(vocab_keys)
        self.key_dtype = vocab_keys.dtype
        self.key_bytes = 4 if self.suffix == "i32" else 8
        self.n_vocab = int(vocab_keys.numel())
        self.first_label = int(first_label)
        self.unique = bool(unique)
        self.capacity = next_pow2(max(64, (4 if self.n_vocab <= (1 << 20) else 2) * self.n_vocab + 1))
        dev = vocab_keys.device
        vk = vocab_keys.contiguous()
        self._vk = vk
        # kept: the head of the (frequency-ordered, duplicate-free) vocabulary is staged in LDS
        self.vocab_keys = vk if unique else None
        # a duplicate-free vocabulary that fits the LDS table in full (include/nvt_hip.h:
        # NVT_ENCODE_RESIDENT_*) is encoded from LDS alone: no global table, no build kernel
        resident = ENCODE_RESIDENT_I32 if self.key_bytes == 4 else ENCODE_RESIDENT_I64
        self.table = self.sentinel_label = None
        self.range_aux, self.range_bits = None, 0
        self.flat_slots = 0
        self.sort_tmp = None
        self._counts = None    # the counts tensor while an internal stream still orders it
        self._src = None       # key-sorted source list of the one-pass ordering, same lifetime
        self.ready = None      # Event recorded behind the sort / build on an internal stream
        self.pending = False   # True until the current stream has been made to wait for it
        # the LDS head of the cache-mode encode, built ONCE per vocabulary by
        # nvt_vocab_finalize_many (include/nvt_hip.h nvt_vocab_col.head_image): only tables whose
        # vocabulary is ordered there (defer_build) get one
        self.head_image = None
        if unique and 0 < self.n_vocab <= resident:
            return
        if ENCODE_HEAD_IMAGE and defer_build and unique and self.key_bytes == 4:
            self.head_image = torch.empty(_lib.ENCODE_HEAD_BYTES, dtype=torch.uint8, device=dev)
        if range_table is not None:
            assert defer_build and unique and self.key_bytes == 4
            self.table, self.range_aux, self.range_bits = range_table
            self.capacity = 0
            self.sentinel_label = torch.empty(1, dtype=torch.int64, device=dev)
            return
        if flat and USE_FLAT_TABLE and defer_build and unique and self.key_bytes == 4:
            # flat range table laid out from the key-sorted source by nvt_vocab_finalize_many
            # (prefix maximum, no random inserts): 2^k >= 2 n slots + n + 64 tail slots
            # home slots at load FLAT_INDEX_LOAD (no power of two: 36 M keys took 2^27 slots = 1 GiB
            # to clear and fill, now 576 MB)
            self.flat_slots = max(64, int(self.n_vocab / FLAT_INDEX_LOAD) + 1)
            self.capacity = self.flat_slots + self.n_vocab + 64
            self.table = torch.empty(self.capacity * 8, dtype=torch.uint8, device=dev)
            self.range_aux = torch.empty(self.FLAT_AUX_WORDS, dtype=torch.int32, device=dev)
            self.sentinel_label = torch.empty(1, dtype=torch.int64, device=dev)
            return
        key = (self.key_bytes, self.capacity)
        nbytes = _ENC_BYTES.get(key)
        if nbytes is None:
            out = C.c_uint64()
            check(self.lib.nvt_encode_table_bytes(self.key_bytes, self.capacity, C.byref(out)))
            nbytes = _ENC_BYTES[key] = out.value
        self.table = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.sentinel_label = torch.empty(1, dtype=torch.int64, device=dev)
        if defer_build:  # built by nvt_vocab_finalize_many, after the vocabulary is ordered
            return
        check(
            getattr(self.lib, f"nvt_encode_build_{self.suffix}")(
                vk.data_ptr() if self.n_vocab else None, self.n_vocab, self.first_label,
                self.table.data_ptr(), self.capacity, self.sentinel_label.data_ptr(),
                1 if unique else 0, stream_ptr(),
            ),
            "nvt_encode_build",
        )

    def fill_vocab_desc(self, d: "_lib.VocabCol", counts: torch.Tensor, max_count: int, src=None):
        """One nvt_vocab_col: order (self vocab keys, counts) in place, then build the table.
        ``src`` = (keys, counts, cls_hist, n_big) of a KEY-SORTED list (range path): the
        vocabulary is then written out of place into (self vocab keys, counts) by one stable
        counting pass that also fills the table."""
        n = self.n_vocab
        if src is not None:
            return self._fill_vocab_desc_sorted(d, counts, max_count, src)
        d.src_keys = d.src_counts = d.cls_hist = d.range_aux = None
        d.n_big = 0
        d.range_nb_log2 = 0
        d.flat_slots = 0
        # the counts are ordered in place on the same internal stream as the keys: they must
        # outlive the hand-off event exactly like keys / table / sort_tmp (a rank that writes
        # no artifacts used to drop its only reference right after the launch)
        self._counts = counts
        d.keys = self._vk.data_ptr()
        d.counts = counts.data_ptr()
        d.n = n
        d.max_count = int(max_count)
        d.key_bytes = self.key_bytes
        d.unique_keys = 1 if self.unique else 0
        small = self.key_bytes == 4 and 2 <= n <= 16384 and 0 < int(max_count) < (1 << 32)
        if n > 1 and not small:
            key = (self.key_bytes, n)
            nbytes = _SORT_BYTES.get(key)
            if nbytes is None:
                out = C.c_uint64()
                check(self.lib.nvt_vocab_sort_tmp_bytes(self.key_bytes, n, C.byref(out)))
                nbytes = _SORT_BYTES[key] = out.value
            self.sort_tmp = torch.empty(nbytes + 16, dtype=torch.uint8, device=self._vk.device)
            d.sort_tmp = self.sort_tmp.data_ptr()
        else:
            d.sort_tmp = None
        d.first_label = self.first_label
        d.table = ptr(self.table)
        d.capacity = self.capacity
        d.sentinel_label = ptr(self.sentinel_label)
        d.head_image = ptr(self.head_image)
        if n > 1 and not small and ASYNC_FINALIZE:
            # ordered on an internal stream: hand-off by event instead of a stream join, so the
            # caller's stream keeps working (fill + normalize, encodes of the small vocabularies)
            # underneath this vocabulary's radix passes and table build
            if self.ready is None:
                self.ready = Event()
            d.ready_event = self.ready.handle
            self.pending = True
        else:
            d.ready_event = None

    def _fill_vocab_desc_sorted(self, d, counts, max_count, src):
        src_keys, src_counts, cls_hist, n_big = src[:4]
        # (a fifth element: int32 positions of the source entries in the vocabulary order -- a
        # multi-GPU fit whose owners labelled their shards; no ordering pass is run then)
        labels = src[4] if len(src) > 4 else None
        if labels is not None:
            assert labels.dtype == torch.int32 and labels.numel() == src_keys.numel()
            labels, n_big = labels.contiguous(), 0
            stat_add("labelled_vocabularies")
        n = self.n_vocab
        assert self.key_bytes == 4 and src_keys.numel() == n and counts.numel() == n
        self._counts = counts
        self._src = (src_keys, src_counts, cls_hist, labels)  # read on an internal stream until `ready`
        d.keys = self._vk.data_ptr()
        d.counts = counts.data_ptr()
        d.n = n
        d.max_count = int(max_count)
        d.key_bytes = 4
        d.unique_keys = 1
        d.src_keys = src_keys.data_ptr()
        d.src_counts = src_counts.data_ptr()
        d.cls_hist = ptr(cls_hist)
        d.src_labels = ptr(labels)
        d.n_big = int(n_big)
        d.range_aux = ptr(self.range_aux)
        d.range_nb_log2 = int(self.range_bits)
        d.flat_slots = int(self.flat_slots)
        key = ("order", n, int(n_big))
        nbytes = _SORT_BYTES.get(key)
        if nbytes is None:
            out = C.c_uint64()
            check(self.lib.nvt_vocab_order_tmp_bytes(n, int(n_big), C.byref(out)))
            nbytes = _SORT_BYTES[key] = out.value
        self.sort_tmp = torch.empty(nbytes + 16, dtype=torch.uint8, device=self._vk.device)
        d.sort_tmp = self.sort_tmp.data_ptr()
        d.first_label = self.first_label
        d.head_image = ptr(self.head_image)
        d.table = ptr(self.table)
        d.capacity = self.capacity
        d.sentinel_label = ptr(self.sentinel_label)
        if ASYNC_FINALIZE:
            if self.ready is None:
                self.ready = Event()
            d.ready_event = self.ready.handle
            self.pending = True
        else:
            d.ready_event = None

    def flat_ok(self) -> bool:
        """Flat range tables only: True when no entry sits further than FLAT_MAX_DISPLACEMENT slots
        from its home slot.  Synchronises (the finalisation of this vocabulary has to be done)."""
        self.wait_ready()
        return int(self.range_aux[self.FLAT_AUX_MAXDISP].item()) & 0xFFFFFFFF <= self.FLAT_MAX_DISPLACEMENT

    def wait_ready(self):
        """Order the CURRENT stream behind this vocabulary's sort / table build (no host
        synchronisation).  Every consumer of the vocabulary or the table calls this first."""
        if self.pending:
            self.ready.wait()
            self.pending = False
            self.sort_tmp = None  # scratch of the finished sort: safe to recycle from here on
            self._counts = None
            self._src = None

    def __del__(self):
        try:
            if getattr(self, "pending", False):
                self.wait_ready()  # the buffers are about to be recycled on this stream
        except Exception:
            pass

    def fill_encode_desc(self, d: "_lib.EncodeCol", keys, valid, null_label, oov_label,
                         num_buckets, out):
        d.keys = keys.data_ptr()
        d.valid = ptr(valid)
        d.n = keys.numel()
        d.table = ptr(self.table)
        d.capacity = self.capacity
        d.sentinel_label = ptr(self.sentinel_label)
        d.null_label = int(null_label)
        d.oov_label = int(oov_label)
        d.num_buckets = int(num_buckets or 0)
        d.key_bytes = self.key_bytes
        d.out_bytes = out.element_size()
        d.out = out.data_ptr()
        d.vocab_keys = ptr(self.vocab_keys) if self.n_vocab else None
        d.n_vocab = self.n_vocab if self.vocab_keys is not None else 0
        d.first_label = self.first_label
        d.range_aux = ptr(self.range_aux)
        d.head_image = ptr(self.head_image)
        if self.pending:
            d.wait_event = self.ready.handle  # nvt_encode_many waits on the launch stream
            self.pending = False
            self.sort_tmp = None
            self._counts = None
            self._src = None
        else:
            d.wait_event = None

    def encode(
        self,
        keys: torch.Tensor,
        valid: Optional[torch.Tensor],
        null_label: int,
        oov_label: int,
        num_buckets: int = 0,
        out_dtype: torch.dtype = torch.int64,
    ) -> torch.Tensor:
        if self.range_aux is not None:  # range tables go through the descriptor entry point
            return encode_many([(self, keys, valid, null_label, oov_label, num_buckets)], out_dtype)[0]
        self.wait_ready()
        if keys.dtype != self.key_dtype:
            keys = keys.to(self.key_dtype)
        keys = aligned(keys)
        if out_dtype not in (torch.int32, torch.int64):
            raise TypeError("Categorify output dtype must be int32 or int64")
        out = torch.empty(keys.numel(), dtype=out_dtype, device=keys.device)
        with _timed(f"encode_{self.suffix}", keys.numel() * (self.key_bytes + out.element_size())):
            check(
                getattr(self.lib, f"nvt_encode_{self.suffix}")(
                    keys.data_ptr(), ptr(valid), keys.numel(), ptr(self.table),
                    self.capacity, ptr(self.sentinel_label), int(null_label),
                    int(oov_label), int(num_buckets or 0), out.data_ptr(), out.element_size(),
                    ptr(self.vocab_keys) if self.n_vocab else None,
                    self.n_vocab if self.vocab_keys is not None else 0, self.first_label,
                    stream_ptr(),
                ),
                "nvt_encode",
            )
        return out


def flat_tables_ok(tabs) -> List[bool]:
    """EncodeTable.flat_ok for several flat range tables with ONE read-back, taken on a side
    stream that waits for nothing but the tables' own finalisation events: work already queued on
    the current stream (the encodes that use these very tables) is neither waited for by the host
    nor held up on the device.  The tables must have been finalised with a ready event."""
    dev = tabs[0].range_aux.device
    side = _check_streams.get(dev.index)
    if side is None:
        side = _check_streams[dev.index] = torch.cuda.Stream(device=dev)
    for t in tabs:
        if t.ready is not None:
            t.ready.wait(stream=side.cuda_stream)
        else:
            side.wait_stream(torch.cuda.current_stream(dev))  # finalised on the caller's stream
    with torch.cuda.stream(side):
        words = torch.stack([t.range_aux[EncodeTable.FLAT_AUX_MAXDISP] for t in tabs]).to(torch.int64)
        vals = read_back(words)
    return [(int(d) & 0xFFFFFFFF) <= EncodeTable.FLAT_MAX_DISPLACEMENT for d in vals]


def encode_many(items, out_dtype: torch.dtype = torch.int64):
    """items: [(EncodeTable, keys, valid, null_label, oov_label, num_buckets)] -> [labels];
    every column of a Categorify.transform enqueued by ONE C call (nvt_encode_many)."""
    if out_dtype not in (torch.int32, torch.int64):
        raise TypeError("Categorify output dtype must be int32 or int64")
    # vocabularies still being ordered on an internal stream go last, largest first (the order
    # nvt_vocab_finalize_many works through them): everything that is ready runs underneath
    order = sorted(range(len(items)),
                   key=lambda i: (1, -items[i][0].n_vocab) if items[i][0].pending else (0, 0))
    descs = (_lib.EncodeCol * max(1, len(items)))()
    outs, keep = [None] * len(items), []
    for d, i in zip(descs, order):
        tab, keys, valid, null_label, oov_label, nb = items[i]
        if keys.dtype != tab.key_dtype:
            keys = keys.to(tab.key_dtype)
        keys = aligned(keys)
        out = torch.empty(keys.numel(), dtype=out_dtype, device=keys.device)
        tab.fill_encode_desc(d, keys, valid, null_label, oov_label, nb, out)
        outs[i] = out
        keep.append(keys)
    if items:
        check(_lib.load().nvt_encode_many(descs, len(items), stream_ptr()), "nvt_encode_many")
    return outs


def hash_bucket(
    keys: torch.Tensor,
    num_buckets: int,
    xor_in: Optional[torch.Tensor] = None,
    want_hash: bool = False,
    want_bucket: bool = True,
    valid: Optional[torch.Tensor] = None,
):
    """(bucket int32 or None, hash64 or None).  hash64 is carried as int64 bits.  Rows whose
    ``valid`` bit is clear hash as key 0 (the slot under a null holds arbitrary bytes)."""
    lib = _lib.load()
    _lib.require_gpu()
    keys = aligned(keys)
    n = keys.numel()
    out = torch.empty(n, dtype=torch.int32, device=keys.device) if want_bucket else None
    xo = torch.empty(n, dtype=torch.int64, device=keys.device) if want_hash else None
    check(
        getattr(lib, f"nvt_hash_bucket_{_key_suffix(keys)}")(
            keys.data_ptr(), ptr(valid), n, int(num_buckets), ptr(out), ptr(xor_in), ptr(xo),
            stream_ptr()
        ),
        "nvt_hash_bucket",
    )
    return out, xo


# --------------------------------------------------------------------------
# continuous columns
# --------------------------------------------------------------------------
_scratch = {}


def _partials(device) -> torch.Tensor:
    # one scratch block per (device, stream): reductions of different operators may be in
    # flight on different streams at the same time
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    if key not in _scratch:
        nbytes = _lib.load().nvt_moments_scratch_bytes()
        _scratch[key] = torch.empty(nbytes // 8, dtype=torch.float64, device=device)
    return _scratch[key]


def moments_accumulate(
    x: torch.Tensor, valid: Optional[torch.Tensor], out3: torch.Tensor, fill: Optional[float] = None
):
    """out3 (float64[3] on device) += {count, sum, sum of squares}."""
    _lib.require_gpu()
    x = aligned(x.view(torch.uint8) if x.dtype == torch.bool else x)
    with _timed("moments", x.numel() * x.element_size()):
        check(
            _lib.load().nvt_moments(
                x.data_ptr(), dtype_code(x.dtype), ptr(valid), x.numel(),
                0 if fill is None else 1, 0.0 if fill is None else float(fill), out3.data_ptr(),
                _partials(x.device).data_ptr(), stream_ptr(),
            ),
            "nvt_moments",
        )


def moments_many(items):
    """items: [(x, valid, fill or None, out3)]: out3 (float64[3] on device) += {count, sum,
    sum of squares}; one launch for all columns (nvt_moments_many)."""
    if not items:
        return
    _lib.require_gpu()
    lib = _lib.load()
    descs = (_lib.MomentsCol * len(items))()
    keep = []
    for d, (x, valid, fill, out3) in zip(descs, items):
        x = aligned(x.view(torch.uint8) if x.dtype == torch.bool else x)
        keep.append(x)
        d.x = x.data_ptr()
        d.valid = ptr(valid)
        d.n = x.numel()
        d.dtype = dtype_code(x.dtype)
        d.has_fill = 0 if fill is None else 1
        d.fill_val = 0.0 if fill is None else float(fill)
        d.out3 = out3.data_ptr()
    dev = keep[0].device
    key = ("many", dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)
    per = lib.nvt_moments_scratch_bytes() // 8
    buf = _scratch.get(key)
    if buf is None or buf.numel() < per * len(items):
        buf = _scratch[key] = torch.empty(per * len(items), dtype=torch.float64, device=dev)
    check(lib.nvt_moments_many(descs, len(items), buf.data_ptr(), stream_ptr()), "nvt_moments_many")


def fill_normalize_many(items):
    """items: [(x, valid, fill, do_norm, shift, scale, out_dtype, want_filled_mask[, moments])] ->
    [(out, filled or None)]; one launch per dtype combination (nvt_fill_normalize_many).
    ``moments``: float64 {count, sum, sum of squares} of the column on the device -- shift / scale
    are then finished from them by the kernel (a fit whose moments are still on their way to the
    host)."""
    if not items:
        return []
    _lib.require_gpu()
    descs = (_lib.FillNormCol * len(items))()
    outs, keep = [], []
    for d, item in zip(descs, items):
        x, valid, fill, do_norm, shift, scale, out_dtype, want_mask = item[:8]
        moments = item[8] if len(item) > 8 else None
        x = aligned(x)
        keep.append(x)
        if moments is not None:
            assert moments.dtype == torch.float64 and moments.is_contiguous() and int(moments.numel()) >= 3
            keep.append(moments)
        d.moments = ptr(moments)
        n = x.numel()
        out = torch.empty(n, dtype=out_dtype, device=x.device)
        filled = torch.empty(n, dtype=torch.uint8, device=x.device) if want_mask else None
        d.x = x.data_ptr()
        d.valid = ptr(valid)
        d.n = n
        d.dtype = dtype_code(x.dtype)
        d.has_fill = 0 if fill is None else 1
        d.fill_val = 0.0 if fill is None else float(fill)
        d.do_norm = 1 if do_norm else 0
        d.out_dtype = dtype_code(out_dtype)
        d.shift = float(shift)
        d.scale = float(scale)
        d.out = out.data_ptr()
        d.filled = ptr(filled)
        outs.append((out, filled.view(torch.bool) if filled is not None else None))
    check(_lib.load().nvt_fill_normalize_many(descs, len(items), stream_ptr()),
          "nvt_fill_normalize_many")
    return outs


def minmax_accumulate(x: torch.Tensor, valid: Optional[torch.Tensor], out2: torch.Tensor, first: bool):
    _lib.require_gpu()
    x = aligned(x)
    check(
        _lib.load().nvt_minmax(
            x.data_ptr(), dtype_code(x.dtype), ptr(valid), x.numel(), 0 if first else 1,
            out2.data_ptr(), _partials(x.device).data_ptr(), stream_ptr(),
        ),
        "nvt_minmax",
    )


def fill_normalize(
    x: torch.Tensor,
    valid: Optional[torch.Tensor],
    fill: Optional[float],
    do_norm: bool,
    shift: float,
    scale: float,
    out_dtype: torch.dtype,
    want_filled_mask: bool = False,
):
    _lib.require_gpu()
    x = aligned(x)
    n = x.numel()
    out = torch.empty(n, dtype=out_dtype, device=x.device)
    filled = torch.empty(n, dtype=torch.uint8, device=x.device) if want_filled_mask else None
    with _timed("fill_normalize", n * (x.element_size() + out.element_size())):
        check(
            _lib.load().nvt_fill_normalize(
                x.data_ptr(), dtype_code(x.dtype), ptr(valid), n, 0 if fill is None else 1,
                0.0 if fill is None else float(fill), 1 if do_norm else 0, float(shift),
                float(scale), out.data_ptr(), dtype_code(out_dtype), ptr(filled), stream_ptr(),
            ),
            "nvt_fill_normalize",
        )
    return out, (filled.view(torch.bool) if filled is not None else None)


def clip_log(x, valid, fill, vmin, vmax, do_log: bool, out_dtype: torch.dtype) -> torch.Tensor:
    """Clip / LogOp pass (optionally consuming a pending FillMissing constant)."""
    _lib.require_gpu()
    x = aligned(x)
    n = x.numel()
    out = torch.empty(n, dtype=out_dtype, device=x.device)
    with _timed("clip_log", n * (x.element_size() + out.element_size())):
        check(
            _lib.load().nvt_clip_log(
                x.data_ptr(), dtype_code(x.dtype), ptr(valid), n, 0 if fill is None else 1,
                0.0 if fill is None else float(fill), 0 if vmin is None else 1,
                0.0 if vmin is None else float(vmin), 0 if vmax is None else 1,
                0.0 if vmax is None else float(vmax), 1 if do_log else 0, out.data_ptr(),
                dtype_code(out_dtype), stream_ptr(),
            ),
            "nvt_clip_log",
        )
    return out


def bucketize(x: torch.Tensor, valid, boundaries: torch.Tensor) -> torch.Tensor:
    """np.digitize(x, boundaries, right=False) as int32; boundaries float64 ascending on device."""
    _lib.require_gpu()
    x = x.contiguous()
    out = torch.empty(x.numel(), dtype=torch.int32, device=x.device)
    check(
        _lib.load().nvt_bucketize(x.data_ptr(), dtype_code(x.dtype), ptr(valid), x.numel(),
                                  ptr(boundaries), int(boundaries.numel()), out.data_ptr(),
                                  stream_ptr()),
        "nvt_bucketize",
    )
    return out


def widen_i64(x: torch.Tensor) -> torch.Tensor:
    if x.dtype == torch.int64:
        return x
    _lib.require_gpu()
    if x.dtype == torch.bool:
        x = x.view(torch.uint8)
    x = x.contiguous()
    out = torch.empty(x.numel(), dtype=torch.int64, device=x.device)
    check(
        _lib.load().nvt_widen_i64(x.data_ptr(), dtype_code(x.dtype), x.numel(), out.data_ptr(),
                                  stream_ptr()),
        "nvt_widen_i64",
    )
    return out


def unpack_bitmap(valid: torch.Tensor, n: int) -> torch.Tensor:
    """Arrow validity bitmap -> bool[n] on the device (torch plumbing for rare paths: combo
    Categorify null rows).  Replaces a device -> host -> device round trip per column."""
    shifts = torch.arange(8, device=valid.device, dtype=torch.uint8)
    bits = (valid[: (n + 7) // 8].unsqueeze(1) >> shifts) & 1
    return bits.reshape(-1)[:n].to(torch.bool)


def popcount(valid: Optional[torch.Tensor], n: int) -> int:
    if valid is None:
        return n
    _lib.require_gpu()
    out = torch.zeros(1, dtype=torch.int64, device=valid.device)
    check(_lib.load().nvt_popcount(valid.data_ptr(), n, out.data_ptr(), stream_ptr()), "nvt_popcount")
    return int(out.item())


# ---- one int32 key column: groupby-aggregate by sorting + flat index -------------------------
SORTED_GROUPBY = os.environ.get("NVT_SORTED_GROUPBY", "1") != "0"
SORTED_GROUPBY_MIN_ROWS = 1 << 15   # below: launch latency, the hash update is as good
SORTED_GROUPBY_MAX_KFOLD = 16
FLAT_INDEX_LOAD = float(os.environ.get("NVT_FLAT_INDEX_LOAD", "0.5"))


# sorted words shared between the aggregates of ONE pass over one partition (Workflow.fit opens
# and closes it around the partition's fit_partition calls; None = no sharing).  Per THREAD: dask
# worker threads of the reference call transform concurrently on different partitions of one
# fitted workflow (SURVEY 8(b) "Threading", categorify.py:1632,1813) -- a pass and what it shares
# belong to the thread that walks the partition.
_TLS = threading.local()


def current_pass_memo():
    """The memo of the pass this THREAD is inside of, or None."""
    return getattr(_TLS, "memo", None)


class pass_memo:
    """``with K.pass_memo():`` -- results that depend only on a partition's columns (the sorted
    (key, row) words of the sort-path groupby) are shared between the operators fitted inside."""

    def __enter__(self):
        self.prev = current_pass_memo()
        _TLS.memo = {}
        return self

    def __exit__(self, *exc):
        _TLS.memo = self.prev
        return False


# Host-side launch sequences (workspaces and scratch buffers cached per (device, stream), the
# read-back mailboxes, the operators' lazily finalised vocabularies and lookup images) are
# serialised by ONE re-entrant lock taken around every operator call (ops/base.py wraps
# transform / fit_partition / fit_end / fit_finalize of every Operator subclass): threads
# interleave at operator granularity, the GPU work stays asynchronous on its streams.
LAUNCH_LOCK = threading.RLock()


def stat_add(name: str, k: int = 1):
    with LAUNCH_LOCK:
        STATS[name] = STATS.get(name, 0) + k


SHARE_SORTED_VALUES = os.environ.get("NVT_SHARE_SORTED_VALUES", "1") != "0"


EXCHANGE_MAX_COLS, EXCHANGE_MAX_CELLS = 64, 4096   # include/nvt_hip.h nvt_exchange_*


MERGE_SORTED_MAX_ROWS = (1 << 26) - 1   # include/nvt_hip.h nvt_count_merge_sorted
MERGE_SORTED_MAX_COLS = 64


IMAGE_LOOKUP_MAX_OUTPUTS = 24   # include/nvt_hip.h: nvt_flat_lookup_image, ncols <= 24


LOOKUP_IMAGES = os.environ.get("NVT_LOOKUP_IMAGES", "1") != "0"
# key -> group through the key directory (nvt_keydir_lookup_image: one 16-byte read per row, laid
# out by two small passes, no read-back) instead of the flat table ("0": nvt_flat_lookup_image)
KEYED_IMAGES = os.environ.get("NVT_KEYED_IMAGES", "1") != "0"
KEYDIR_LOAD = float(os.environ.get("NVT_KEYDIR_LOAD", "1.0"))   # keys per directory bucket
# records written whole by one launch over every operator's range (nvt_image_build); "0": a launch
# per operator (nvt_jg_image / nvt_te_image)
ONE_PASS_IMAGES = os.environ.get("NVT_ONE_PASS_IMAGES", "1") != "0"
# Workflow.fit ends by enqueueing the lookup images of its groupby operators (FlatIndex.prepare_image)
# instead of leaving them to the first transform; "0": built lazily by the first lookup.
EAGER_IMAGES = os.environ.get("NVT_EAGER_IMAGES", "1") != "0"

# ---- the rest of the driver lives beside this file; the facade re-exports it ---------------------
from .kernels_groupby import (  # noqa: E402,F401
    GroupbyTable, sorted_groupby_eligible, sorted_groupby, flat_index_for, te_apply_folds, _order_ws, order_rows, seg_aggregate, gather, te_apply,
)
from .kernels_lookup import (  # noqa: E402,F401
    FlatIndex, _value_bits, LookupConsumer, image_pack, JG_KINDS, jg_image, jg_image_part, te_image_part, te_image,
)
from .kernels_exchange import (  # noqa: E402,F401
    ExchangeBatch, exchange_unpack, merge_counts_sorted,
)
