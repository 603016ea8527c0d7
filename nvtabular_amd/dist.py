"""Multi-GPU merge of fit statistics: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference moves per-partition groupby frames between dask workers with a
tree reduce over TCP/UCX and a disk round trip (categorify.py:1423-1529).  Here
each rank keeps its partial tables in HBM and the merge is ONE exchange per fit:

  Categorify / JoinGroupby / TargetEncoding fit
      owner(key) = key range r of `world` equal ranges of the column's global key span
                   (monotone in the key: a key-sorted list splits into contiguous slices, the
                   owners' shards concatenate to a key-sorted union; groupby rows: h32 % world)
      all-to-all(v) of the (key, count[, sums...]) rows to their owners   <- xGMI, all 7 links busy
      owner-side merge: a merge tree over the G key-ordered runs of every column
                   (nvt_merge_sorted_many; lists that arrive unordered: one sort,
                   nvt_count_merge_sorted; groupby rows: nvt_gb_merge)
      all-gather of the merged shards -> every rank holds the full table and runs the
      same deterministic finalisation (one-pass ordering, thresholds), so vocabularies are
      identical; the class histograms the ordering needs are summed over the owners
  Normalize.fit / target means
      all-reduce(sum) of the 3*K float64 moment vector (latency-bound, 312 B for K=13)

transform needs no communication (tables are replicated).

The choreography below is backend-agnostic; the two device-specific steps
(owner hashing, owner-side merge) are injected so the world_size-2 gloo tests can
drive the same code path with host implementations from tests/.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as td


_local_only = 0


class local_only:
    """Within this block the process behaves as a single-rank job although a process group
    exists: a fit made by ONE rank alone (bench.py's parity leg on rank 0) must not enter the
    collectives its peers are not taking part in."""

    def __enter__(self):
        global _local_only
        _local_only += 1
        return self

    def __exit__(self, *exc):
        global _local_only
        _local_only -= 1
        return False


def world_size() -> int:
    if _local_only:
        return 1
    return td.get_world_size() if td.is_available() and td.is_initialized() else 1


def rank() -> int:
    if _local_only:
        return 0
    return td.get_rank() if td.is_available() and td.is_initialized() else 0


def _backend() -> str:
    return td.get_backend() if world_size() > 1 else "none"


# --------------------------------------------------------------------------
# collectives with a gloo-safe fallback
# --------------------------------------------------------------------------
def _staged() -> bool:
    """gloo moves host memory only: with gloo (CPU tests, or the single-GPU end-to-end rehearsal
    `NVT_BENCH_BACKEND=gloo`) device tensors go through the host around every collective."""
    return _backend() == "gloo"


# bytes this rank hands to / receives from every collective form since the last reset_traffic():
# host arithmetic on shapes only (always on; the bench puts it into every N > 1 line)
TRAFFIC = {}


def _count_traffic(name: str, sent: int, received: int):
    rec = TRAFFIC.setdefault(name, {"calls": 0, "bytes_sent": 0, "bytes_received": 0})
    rec["calls"] += 1
    rec["bytes_sent"] += int(sent)
    rec["bytes_received"] += int(received)


def reset_traffic():
    TRAFFIC.clear()


def _nbytes(t: torch.Tensor) -> int:
    return int(t.numel()) * t.element_size()


def _all_reduce(t: torch.Tensor, op=td.ReduceOp.SUM) -> torch.Tensor:
    _count_traffic("all_reduce", _nbytes(t), _nbytes(t))
    if _staged() and t.is_cuda:
        h = t.cpu()
        td.all_reduce(h, op=op)
        t.copy_(h)
    else:
        td.all_reduce(t, op=op)
    return t


def _all_gather_same(t: torch.Tensor) -> List[torch.Tensor]:
    """Every rank's tensor of identical shape, rank order."""
    G = world_size()
    _count_traffic("all_gather(equal shapes)", _nbytes(t), _nbytes(t) * G)
    if _staged() and t.is_cuda:
        h = t.cpu()
        bufs = [torch.empty_like(h) for _ in range(G)]
        td.all_gather(bufs, h)
        return [b.to(t.device) for b in bufs]
    bufs = [torch.empty_like(t) for _ in range(G)]
    td.all_gather(bufs, t.contiguous())
    return bufs


def all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    if world_size() > 1:
        _all_reduce(t, td.ReduceOp.SUM)
    return t


def _nan_reduce(t: torch.Tensor, op) -> torch.Tensor:
    if world_size() == 1:
        return t
    big = float("inf") if op == td.ReduceOp.MIN else float("-inf")
    x = torch.where(torch.isnan(t), torch.full_like(t, big), t)
    _all_reduce(x, op)
    return torch.where(torch.isinf(x), torch.full_like(x, float("nan")), x)


def all_reduce_min(t):
    return _nan_reduce(t, td.ReduceOp.MIN)


def all_reduce_max(t):
    return _nan_reduce(t, td.ReduceOp.MAX)


def _all_to_all_counts(send_counts: torch.Tensor) -> torch.Tensor:
    """int64[G] rows this rank sends to each peer -> int64[G] rows it receives."""
    G = world_size()
    recv = torch.empty_like(send_counts)
    if _backend() == "nccl":
        _count_traffic("all_to_all_single(counts)", _nbytes(send_counts), _nbytes(send_counts))
        td.all_to_all_single(recv, send_counts)
    else:
        r = rank()
        recv = torch.stack([g[r] for g in _all_gather_same(send_counts)])
    return recv


def _all_to_all_v(send: torch.Tensor, send_counts: List[int], recv_counts: List[int]) -> torch.Tensor:
    """Variable-size all-to-all along dim 0 of a tensor already grouped by destination rank."""
    G = world_size()
    out = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype,
                      device=send.device)
    _count_traffic("all_to_all_single(uneven)", _nbytes(send), _nbytes(out))
    if _backend() == "nccl":
        td.all_to_all_single(out, send.contiguous(), output_split_sizes=recv_counts,
                             input_split_sizes=send_counts)
        return out
    # gloo: pairwise exchange through host memory (tests / single-GPU rehearsal only)
    dev = send.device
    if send.is_cuda:
        send = send.cpu()
        out = torch.empty(out.shape, dtype=out.dtype)
    r = rank()
    s_off = [0]
    for c in send_counts:
        s_off.append(s_off[-1] + c)
    r_off = [0]
    for c in recv_counts:
        r_off.append(r_off[-1] + c)
    out[r_off[r] : r_off[r + 1]] = send[s_off[r] : s_off[r + 1]]
    reqs = []
    for peer in range(G):
        if peer == r:
            continue
        if send_counts[peer]:
            reqs.append(td.isend(send[s_off[peer] : s_off[peer + 1]].contiguous(), peer))
    for peer in range(G):
        if peer == r or not recv_counts[peer]:
            continue
        buf = torch.empty((recv_counts[peer],) + tuple(send.shape[1:]), dtype=send.dtype)
        td.recv(buf, peer)
        out[r_off[peer] : r_off[peer + 1]] = buf
    for q in reqs:
        q.wait()
    return out.to(dev)


def _all_gather_v(t: torch.Tensor, sizes: Optional[List[int]] = None) -> torch.Tensor:
    """Concatenate every rank's tensor along dim 0 (variable length), rank order.
    ``sizes`` = the per-rank lengths when the caller already knows them."""
    G = world_size()
    if sizes is None:
        n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
        sizes = [int(s.item()) for s in _all_gather_same(n)]
    row = (int(t[0].numel()) if t.dim() > 1 and t.shape[0] else 1) * t.element_size()
    _count_traffic("all_gather_v", _nbytes(t) * (G - 1), sum(sizes) * row)
    if _backend() == "nccl":
        # exact sizes, no padding to the longest shard (key-range owners are not balanced):
        # an all-to-all(v) in which every rank sends its whole tensor to every peer moves
        # exactly the bytes of an all-gather(v)
        out = torch.empty((sum(sizes),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        n = int(t.shape[0])
        send = t.contiguous().repeat((G,) + (1,) * (t.dim() - 1))
        td.all_to_all_single(out, send, output_split_sizes=list(sizes), input_split_sizes=[n] * G)
        return out
    m = max(sizes) if sizes else 0
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = _all_gather_same(pad)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])


def _pack64(columns: Sequence[torch.Tensor]) -> torch.Tensor:
    """Columns of one table as ONE int64 [rows, ncols] matrix (float64 bit-cast, narrower integer
    types widened), so that a table travels in one collective instead of one per column."""
    cols = []
    for c in columns:
        if c.dtype.is_floating_point:
            # any float travels as the bits of its float64 value (exact for float32 / float16;
            # a plain .to(int64) would truncate it)
            cols.append(c.to(torch.float64).contiguous().view(torch.int64))
        elif c.dtype in (torch.int64, torch.int32, torch.int16, torch.int8, torch.uint8, torch.bool):
            cols.append(c.to(torch.int64))
        else:
            raise TypeError(f"exchange_rows / gather_rows: unsupported column dtype {c.dtype}")
    if not cols:
        return torch.empty((0, 0), dtype=torch.int64)
    return torch.stack(cols, dim=1)


def _unpack64(mat: torch.Tensor, like: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    out = []
    for j, c in enumerate(like):
        col = mat[:, j].contiguous()
        out.append(col.view(torch.float64).to(c.dtype) if c.dtype.is_floating_point else col.to(c.dtype))
    return out


def exchange_rows(columns: Sequence[torch.Tensor], owner: torch.Tensor) -> List[torch.Tensor]:
    """Send row i of every column to rank owner[i]; returns the received columns.  ONE
    all-to-all(v) for the whole table (it used to be one per column: >= 8 latency-bound
    collectives per JoinGroupby / TargetEncoding table)."""
    G = world_size()
    owner64 = owner.to(torch.int64)
    if owner.is_cuda:
        from . import kernels as K

        order = K.order_rows(int(owner.numel()), owner.device, gid=owner64, ngroups=G) & 0xFFFFFFFF
    else:
        order = torch.argsort(owner64, stable=True)
    send_counts_t = torch.bincount(owner64, minlength=G).to(torch.int64)
    recv_counts_t = _all_to_all_counts(send_counts_t)
    sc, rc = send_counts_t.cpu().tolist(), recv_counts_t.cpu().tolist()
    mat = _pack64(columns)[order].contiguous()
    return _unpack64(_all_to_all_v(mat, sc, rc), columns)


def gather_rows(columns: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """Every rank's rows of a table, concatenated in rank order: ONE all-gather(v)."""
    return _unpack64(_all_gather_v(_pack64(columns)), columns)


# --------------------------------------------------------------------------
# device-specific steps (HIP by default; tests inject host versions)
# --------------------------------------------------------------------------
def _hip_owner(keys_list: Sequence[torch.Tensor], G: int) -> torch.Tensor:
    from . import kernels as K

    acc = None
    owner = None
    for i, k in enumerate(keys_list):
        last = i == len(keys_list) - 1
        owner, acc = K.hash_bucket(k, G, xor_in=acc, want_hash=not last, want_bucket=last)
    return owner


def _hip_merge_counts(keys: torch.Tensor, counts: torch.Tensor):
    from . import kernels as K

    k, c, _, _ = K.dense_count(keys, None, counts, hint=int(keys.numel()))
    return k, c


def _hip_merge_counts_many(parts):
    """Owner-side merge of every column's received (key, count) rows: ONE batched launch and ONE
    read-back for all columns (the per-column form synchronised 26 times per Criteo fit)."""
    from . import kernels as K

    jobs = [K.DenseCountJob(k, None, c, hint=int(k.numel())) for k, c in parts]
    return [(k, c) for k, c, _, _ in K.dense_count_many(jobs)]


_owner_fn: Callable = _hip_owner
_merge_counts_fn: Callable = _hip_merge_counts
_merge_counts_many_fn: Optional[Callable] = _hip_merge_counts_many


def set_backend_fns(owner_fn=None, merge_counts_fn=None):
    """Test hook: replace the HIP owner-hash / owner-merge steps (gloo CPU tests)."""
    global _owner_fn, _merge_counts_fn, _merge_counts_many_fn
    _owner_fn = owner_fn or _hip_owner
    _merge_counts_fn = merge_counts_fn or _hip_merge_counts
    # an injected per-column merge (host stand-in of the CPU tests) replaces the batched one too
    _merge_counts_many_fn = None if merge_counts_fn else _hip_merge_counts_many


# --------------------------------------------------------------------------
# fit-statistics merges
# --------------------------------------------------------------------------
def merge_counts(keys: torch.Tensor, counts: torch.Tensor, nulls: int):
    """Global (key -> count) table from per-rank tables; identical on every rank."""
    if world_size() == 1:
        return keys, counts, nulls
    k, c, sc, _ = merge_counts_many([(keys, counts, [nulls])])[0]
    return k, c, sc[0]


def _range_owner(k64: torch.Tensor, lo: int, hi: int, G: int) -> torch.Tensor:
    """Owner rank of every key by KEY RANGE: rank r owns [lo + r * w, lo + (r + 1) * w) with
    w = ceil((hi - lo + 1) / G) -- monotone in the key, so the owners' key-sorted shards,
    concatenated in rank order, are one key-sorted list.  (Plain integer arithmetic on int64
    tensors: O(#distinct keys) plumbing, runs on host tensors in the gloo tests.)"""
    span = hi - lo + 1
    width = -(-span // G)
    # (k - lo) can exceed int64 for int64 keys spanning the whole range: halve both first
    if span >= (1 << 62):
        return (((k64 >> 1) - (lo >> 1)) // max(width >> 1, 1)).clamp_(0, G - 1)
    return ((k64 - lo) // width).clamp_(0, G - 1)


TIMING = None  # NVT_DIST_TIMING=1: dict section -> seconds (device-synchronised: diagnostic only)
if __import__("os").environ.get("NVT_DIST_TIMING"):
    TIMING = {}


_last_mark = [0.0]


def enable_timing(on: bool = True):
    """Section timing for ONE diagnostic fit (device-synchronised at every mark, so never inside a
    timed region): `dist.TIMING` = {section: seconds}.  bench.py runs one such fit per N > 1 line."""
    global TIMING
    TIMING = {} if on else None


def _mark(name=None):
    """Diagnostic timing (NVT_DIST_TIMING=1 / enable_timing): seconds since the previous mark go to
    TIMING[name]."""
    if TIMING is None:
        return
    import time

    if torch.cuda.is_available():
        torch.cuda.synchronize()
    now = time.perf_counter()
    if name is not None:
        TIMING[name] = TIMING.get(name, 0.0) + now - _last_mark[0]
    _last_mark[0] = now


PACK_COUNT_ROWS = True  # (tests switch it off to drive the two-word format with int32 keys)
STATS = {"packed_exchanges": 0, "plain_exchanges": 0, "sorted_merges": 0}  # diagnostics
MERGE_BY_SORTING = True
HIP_EXCHANGE = True  # batched nvt_exchange_* launches around the collectives (int32 keys on the GPU)
# Key-sorted lists (what the range / sort paths and the partition merge produce) travel as
# contiguous slices in key order, and the owner MERGES the G sorted runs of a column
# (nvt_merge_sorted_many) instead of sorting everything it received; lists that are not sorted
# (the LDS-resident counting paths: a few thousand keys) are sorted on the sender first.
ORDERED_EXCHANGE = True
# ... and the owners also ORDER their shards: with the class histograms of every owner each rank
# computes the vocabulary positions ("count descending, key ascending") of its own entries
# (nvt_vocab_label_shard; the few entries with count >= 255 are gathered and sorted exactly by
# everyone) and the labels travel with the merged rows.  Every rank then lays its tables out from
# (key, label) without ordering the union -- the part of a fit that every rank used to repeat for
# ALL G shards (~6 ms of kernels for the 8-rank union of the Criteo vocabularies).
DISTRIBUTED_ORDER = True
STATS["distributed_orders"] = 0
SMALL_SORT_MAX = 1 << 18   # entries of unsorted lists one tagged sort may take on the sender
STATS["ordered_exchanges"] = 0


def _stable_order(words: torch.Tensor) -> torch.Tensor:
    """Indices that sort non-negative int64 `words` ascending, stable.  On the GPU: the library's
    radix order (nvt_sort_key_u64 + nvt_order_rows, two 32-bit passes -- the sort of the Groupby
    operator), not torch.sort; the host stand-in serves the gloo tests on CPU tensors."""
    if words.is_cuda:
        from . import kernels as K

        perm = K.order_rows(int(words.numel()), words.device, [(words.contiguous(), None, True)])
        return perm & 0xFFFFFFFF
    return torch.argsort(words, stable=True)


def _sort_unsorted_lists(tables, sorted_by_key):
    """The lists flagged unsorted ordered by key with ONE sort of (column << 32 | biased key)
    words; None when they hold more than SMALL_SORT_MAX entries (the caller keeps the unordered
    exchange)."""
    todo = [j for j, (k, _, _) in enumerate(tables) if not sorted_by_key[j] and int(k.numel()) > 1]
    if not todo:
        return tables
    lens = [int(tables[j][0].numel()) for j in todo]
    if sum(lens) > SMALL_SORT_MAX:
        return None
    words = torch.cat([(tables[j][0].to(torch.int64) + (1 << 31)) | (i << 32) for i, j in enumerate(todo)])
    cnts = torch.cat([tables[j][1].to(torch.int64) for j in todo])
    order = _stable_order(words)
    words, cnts = words[order], cnts[order]
    keys = ((words & 0xFFFFFFFF) - (1 << 31)).to(torch.int32)
    out, at = list(tables), 0
    for j, n in zip(todo, lens):
        out[j] = (keys[at:at + n], cnts[at:at + n], tables[j][2])
        at += n
    return out


def _pack_kc(k64: torch.Tensor, c64: torch.Tensor) -> torch.Tensor:
    """(int32-ranged key, count < 2^31) -> one int64 word: count in the high half."""
    return (c64 << 32) | (k64 & 0xFFFFFFFF)


def _unpack_kc(words: torch.Tensor, key_dtype):
    keys = ((words << 32) >> 32).to(key_dtype)  # arithmetic shifts: the low half, sign-extended
    return keys.contiguous(), (words >> 32).contiguous()


def _sort_by_key_default(keys, counts):
    if keys.is_cuda:
        from . import kernels as K

        return K.sort_by_key(keys, counts)
    order = torch.argsort(keys, stable=True)  # host stand-in (gloo tests)
    return keys[order].contiguous(), counts[order].contiguous()


def _class_hist_default(counts):
    if counts.is_cuda:
        from . import kernels as K

        return K.class_hist(counts)
    return torch.bincount(counts.clamp(max=255), minlength=256).to(torch.int32)


_sort_by_key_fn: Callable = _sort_by_key_default
_class_hist_fn: Callable = _class_hist_default


def merge_counts_many(tables, sorted_by_key=None, rows_bound=None):
    """ONE exchange for all the (key -> count) tables of a fit.

    ``sorted_by_key[j]``: the list of column j is in ascending key order without duplicates on
    THIS rank (default: unknown = no).  When every column is (or, for short lists, can cheaply be
    made) key-sorted on every rank, groups travel in key order and the owners merge sorted runs
    (ORDERED_EXCHANGE; the decision rides on the MAX all-reduce of the key ranges).
    ``rows_bound[j]``: an upper bound of the sum of the counts of column j on this rank (the rows
    it fitted); with it and sorted lists the key ranges are the first and last keys and no pass
    over the lists is needed for them.

    ``tables`` = [(keys, counts, scalars)] per column, ``scalars`` a list of ints that are
    summed over the ranks (null rows, valid rows, ...).  Every column's rows travel in the
    same collectives, tagged by column:

        all-reduce of the per-column key ranges   (1 collective: min / max of the keys)
        owner(key) = key range r of G equal ranges per column (monotone in the key)
        rows sorted by (owner, column)            -> [G x ncol] send-count matrix
        all-to-all of the count matrix            (1)
        all-to-all(v) of the (key, count) rows    (1; one int64 word per row for int32 keys
                                                   and counts < 2^31, else int64 pairs)
        owner-side merge: ONE sort of the received rows by (column, key) + segmented sum
        (nvt_count_merge_sorted; else weighted dense count per column, then ordered BY KEY)
        all-gather of the [ncol] merged lengths   (1)
        all-gather(v) of the merged rows          (1)
        all-reduce of the scalars                 (1)

    -- 6 collectives per fit however many columns there are.  (Ordered exchange + distributed
    ordering, the default for key-sorted int32 lists on the GPU: the lengths and the scalars ride
    on ONE all-gather of the owners' class histograms, then an all-gather(v) of the entries with
    count >= 255, of the merged rows and of their labels: 7 collectives.)  Because owners hold key RANGES
    and order their shards by key, the gathered list of a column is key-sorted on every rank:
    the vocabulary order (count desc, key asc) is then ONE stable counting pass per rank
    (nvt_vocab_col.src_keys) instead of a 7-pass radix sort of the union on every rank -- the
    part of the finalisation that grew with the number of GPUs.  (The reference partitions the
    uniques for the same reason: split_out, categorify.py:152-160,1214-1270.)
    Returns [(keys, counts, summed_scalars, info)], identical on every rank; info =
    dict(sorted_by_key=True, cls_hist=int32[256], n_big) for the ordering pass.  Keys come back
    in the dtype of the input (columns whose local table is empty everywhere: int64)."""
    G = world_size()
    if G == 1:
        return [(k, c, list(sc), None) for k, c, sc in tables]
    ncol = len(tables)
    _mark()
    dev = tables[0][0].device
    dtypes = [k.dtype for k, _, _ in tables]
    lens = [int(k.numel()) for k, _, _ in tables]
    # A table without entries has no key dtype of its own (Categorify.fit_end passes int64 empties
    # for every column on a rank that received no partition).  The wire format (`packed`) and the
    # returned key dtype must be the SAME decision on every rank, so they are derived from a
    # "some rank holds non-int32 keys" flag that travels in the MAX all-reduce below -- never
    # from the rank-local dtypes (a rank with empties used to build two-word rows while its
    # peers sent one-word rows: the collectives then disagreed on byte counts).
    wide_local = [int(n > 0 and dt != torch.int32) for dt, n in zip(dtypes, lens)]
    if not any(wide_local):
        tables = [(k if k.dtype == torch.int32 else k.to(torch.int32), c, sc) for k, c, sc in tables]
    # key order of the local lists (the ordered exchange needs it on every rank)
    flags = list(sorted_by_key) if sorted_by_key is not None else [False] * ncol
    flags = [bool(f) or n <= 1 for f, n in zip(flags, lens)]
    unsorted_local = 1
    if (ORDERED_EXCHANGE and dev.type == "cuda" and HIP_EXCHANGE and not any(wide_local)
            and _merge_counts_many_fn is _hip_merge_counts_many):
        made = _sort_unsorted_lists(tables, flags)
        if made is not None:
            tables, unsorted_local = made, 0
    # ---- global key range per column ---------------------------------------------------
    big = torch.iinfo(torch.int64).max
    # device path: all columns int32 on the GPU -- every step around the collectives is ONE
    # batched launch (nvt_exchange_*, kernels.ExchangeBatch) instead of ~8 torch kernels per column
    xb = None
    if (dev.type == "cuda" and HIP_EXCHANGE and _merge_counts_many_fn is _hip_merge_counts_many
            and not any(wide_local)):
        from . import kernels as K

        if ncol <= K.EXCHANGE_MAX_COLS and G * ncol <= K.EXCHANGE_MAX_CELLS:
            xb = K.ExchangeBatch([(k, c.to(torch.int64)) for k, c, _ in tables])
    # (-min, max, rows counted on this rank, holds non-int32 keys, holds lists that are not in key
    # order): one MAX reduce
    rng = torch.empty(ncol, 5, dtype=torch.int64, device=dev)
    rng[:, 3] = torch.tensor(wide_local, dtype=torch.int64).to(dev)
    rng[:, 4] = unsorted_local
    if xb is not None:
        if unsorted_local == 0 and rows_bound is not None:
            rng[:, :3] = xb.ranges_sorted()
            rng[:, 2] = torch.tensor([int(b) for b in rows_bound], dtype=torch.int64).to(dev)
        else:
            rng[:, :3] = xb.ranges()
        k64s = None
    else:
        k64s = [k.to(torch.int64) for k, _, _ in tables]
        for j, (k, (_, c, _)) in enumerate(zip(k64s, tables)):
            if lens[j]:
                rng[j, 0] = -(k.min().clamp(min=-big))
                rng[j, 1] = k.max()
                rng[j, 2] = c.sum()
            else:
                rng[j, 0] = -big
                rng[j, 1] = -big
                rng[j, 2] = 0
    _all_reduce(rng, td.ReduceOp.MAX)
    rng_h = rng.cpu().tolist()
    _mark("ranges")
    # the key dtype every rank returns: int64 when ANY rank holds non-int32 keys of the column,
    # int32 when some rank holds keys and all of them are int32, the input's when nobody does
    wide = [r[3] > 0 for r in rng_h]
    dtypes = [torch.int64 if w else (torch.int32 if r[1] != -big else dt)
              for w, r, dt in zip(wide, rng_h, dtypes)]
    # int32 keys and no merged count that can reach 2^31 (G times the largest per-rank total
    # bounds it): a (key, count) row travels as ONE int64 word instead of two -- half the bytes
    # of the all-to-all and of the all-gather, the part of a fit that grows with the ranks
    packed = (not any(wide)
              and G * max(r[2] for r in rng_h) < (1 << 31) and PACK_COUNT_ROWS)
    STATS["packed_exchanges" if packed else "plain_exchanges"] += 1
    if xb is not None and not packed:  # (counts too large for one word: the general path)
        xb = None
        k64s = [k.to(torch.int64) for k, _, _ in tables]
    ordered = xb is not None and not any(r[4] > 0 for r in rng_h)   # the same on every rank
    if xb is not None:
        send_mat, recv_mat, send_h, recv_h, recv = _exchange_rows_hip(xb, rng_h, G, ncol, ordered)
    else:
        send_mat, recv_mat, send_h, recv_h, recv = _exchange_rows_torch(
            tables, k64s, lens, rng_h, G, ncol, dev, packed)
    _mark("all_to_all")
    # ---- owner-side merge, column by column, then key order -----------------------------
    off = torch.zeros(G * ncol + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(recv_h.reshape(-1), 0)  # received layout: source-major, column-minor
    off = off.tolist()
    sorted_merge, results, packed_all, own_hist, own_labels, pre = None, {}, None, None, None, None
    if ordered:
        # G key-ordered runs per column: merge tree (one launch pair per level for all columns)
        sorted_merge, packed_all, packed_len, own_hist, own_labels, pre = _merge_sorted_runs(
            recv, off, G, ncol, [sc for _, _, sc in tables])
        STATS["ordered_exchanges"] += 1
    elif packed and recv.is_cuda and _merge_counts_many_fn is _hip_merge_counts_many and MERGE_BY_SORTING:
        from . import kernels as K

        if 0 < recv.numel() <= K.MERGE_SORTED_MAX_ROWS and ncol <= K.MERGE_SORTED_MAX_COLS:
            # ONE sort of all received rows by (column, key) + a segmented sum: the merged lists
            # come out ordered by key (no hash tables, no per-column sort afterwards)
            sorted_merge, packed_all, packed_len = K.merge_counts_sorted(recv, off, ncol, want_packed=True)
            STATS["sorted_merges"] += 1
    return _merge_and_replicate(tables, dtypes, G, ncol, dev, packed, recv, off, sorted_merge,
                                packed_all, packed_len if sorted_merge is not None else None,
                                xb is not None, own_hist, own_labels, pre)


SMALL_MERGE_MAX = 1 << 16  # received entries of a column up to which its runs are merged by sorting


def _merge_sorted_runs(recv, off, G, ncol, scalars=None):
    """Owner side of the ordered exchange: the received words lie in (source, column) segments,
    every segment in key order.  -> (True, the merged rows of all columns as one array of words,
    column after column, their lengths, int64[ncol, 256] class histogram of this owner's share).
    Columns with many entries: merge tree over their G runs (one launch pair per level for all of
    them); columns with a few thousand entries (half of a Criteo schema, and 7 of every 14 pairs
    of the tree -- whose cost for them is the Python that describes the pairs): ONE sort of
    (column << 32 | key) words of all of them + a segmented sum."""
    from . import kernels as K

    dev = recv.device
    tot = [sum(off[src * ncol + j + 1] - off[src * ncol + j] for src in range(G)) for j in range(ncol)]
    small = [j for j in range(ncol) if 0 < tot[j] <= SMALL_MERGE_MAX]
    big = [j for j in range(ncol) if tot[j] > SMALL_MERGE_MAX]
    dst, runs, at = [0] * (G * ncol), {j: [] for j in big}, 0
    for j in small:      # the small columns first, their runs back to back: one slice holds them all
        for src in range(G):
            sgm = src * ncol + j
            dst[sgm] = at
            at += off[sgm + 1] - off[sgm]
    n_small = at
    at = (at + 3) // 4 * 4
    for j in big:
        for src in range(G):
            sgm = src * ncol + j
            n = off[sgm + 1] - off[sgm]
            dst[sgm] = at
            if n:
                runs[j].append((at, n))
            at += (n + 3) // 4 * 4   # every run starts on a 16-byte boundary of both arrays
    keys_all, cnts_all = K.exchange_unpack(recv, off, dst, at)
    empty = (torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.int64, device=dev))
    merged = [empty] * ncol
    if big:
        for j, m in zip(big, K.merge_sorted_tree([[(keys_all[a:a + n], cnts_all[a:a + n]) for a, n in runs[j]]
                                                  for j in big])):
            merged[j] = m
    if small:
        # (column, key) groups of all the small columns by ONE tagged sort + segmented sum of the
        # library (nvt_count_merge_sorted: the owner-side merge of the unordered exchange)
        seg = [0]
        for j in small:
            seg.append(seg[-1] + tot[j])
        rows = (cnts_all[:n_small] << 32) | (keys_all[:n_small].to(torch.int64) & 0xFFFFFFFF)
        for j, m in zip(small, K.merge_counts_sorted(rows, seg, len(small))):
            merged[j] = m
    lens = [int(k.numel()) for k, _ in merged]
    hist32 = torch.zeros(ncol, 256, dtype=torch.int32, device=dev)
    for j, (_, c) in enumerate(merged):
        if lens[j]:
            K.class_hist(c, out=hist32[j])
    hist = hist32.to(torch.int64) & 0xFFFFFFFF
    labels, pre = (_label_own_shards(merged, lens, hist32, G, ncol, dev, scalars)
                   if DISTRIBUTED_ORDER else (None, None))
    if sum(lens) == 0:
        return True, torch.empty(0, dtype=torch.int64, device=dev), lens, hist, labels, pre
    xm = K.ExchangeBatch([(k, c) for k, c in merged])
    starts = torch.zeros(1, ncol, dtype=torch.int64)
    starts[0, 1:] = torch.cumsum(torch.tensor(lens[:-1], dtype=torch.int64), 0)
    packed_all = xm.pack_ordered([0] * ncol, [1] * ncol, 1, starts.to(dev),
                                 torch.zeros(1, ncol, dtype=torch.int64, device=dev))
    return True, packed_all, lens, hist, labels, pre


def _class_base_diff(hists: torch.Tensor, r: int) -> torch.Tensor:
    """The class bases owner r hands to nvt_vocab_label_shard, as the difference array the kernel
    integrates.  hists int64[G, ncol, 256] = every owner's histogram of min(count, 255).
    base(c), c = 1 .. 254 = entries of the union in the classes 255 .. c + 1 (they come first in
    "count descending") + entries of class c on the owners in front of r (key order = owner
    order).  The kernel takes an exclusive prefix over the classes 255, 254, ...: the value at
    255 is base(254), at c (2 <= c <= 254) it is base(c - 1) - base(c) modulo 2^32.  -> int32[ncol, 256]"""
    ncol = hists.shape[1]
    H, P = hists.sum(0), hists[:r].sum(0)
    Hc = H[:, 1:255]
    above = torch.flip(torch.cumsum(torch.flip(Hc, [1]), 1), [1]) - Hc
    base = H[:, 255:256] + above + P[:, 1:255]                      # [ncol, 254], column c - 1
    diff = torch.zeros(ncol, 256, dtype=torch.int64)
    diff[:, 255] = base[:, 253]
    diff[:, 2:255] = base[:, 0:253] - base[:, 1:254]                # base(c - 1) - base(c)
    diff &= 0xFFFFFFFF
    return torch.where(diff >= (1 << 31), diff - (1 << 32), diff).to(torch.int32)


def _label_own_shards(merged, lens, hist32, G, ncol, dev, scalars=None):
    """Vocabulary positions (0-based, "count descending, key ascending" over the UNION of all
    owners' shards) of this owner's entries: int32, column after column like the packed rows.
    Collectives: one all-gather of the class histograms [ncol, 256] -- the merged lengths and
    the caller's scalars ride on it, so the replicate step needs neither its own length all-gather
    nor the scalar all-reduce --, one all-gather(v) of the entries with count >= 255 (sizes known
    from the histograms).  Returns (labels, dict(all_len [G, ncol], scal [ncol][nsc] summed over
    the ranks, hist [ncol, 256] of the union))."""
    from . import kernels as K

    r = rank()
    nsc = max([len(sc) for sc in scalars] + [0]) if scalars is not None else 0
    meta = torch.zeros(ncol, 257 + nsc, dtype=torch.int64)
    meta[:, 256] = torch.tensor(lens, dtype=torch.int64)
    if nsc:
        meta[:, 257:] = torch.tensor([list(sc) + [0] * (nsc - len(sc)) for sc in scalars], dtype=torch.int64)
    meta = meta.to(dev)
    meta[:, :256] = hist32.to(torch.int64) & 0xFFFFFFFF
    gathered = torch.stack(_all_gather_same(meta)).cpu()          # [G, ncol, 257 + nsc]
    hists = gathered[:, :, :256]
    pre = dict(all_len=gathered[:, :, 256].contiguous(), scal=gathered[:, :, 257:].sum(0).tolist(),
               hist=hists.sum(0), nsc=nsc)
    diff = _class_base_diff(hists, r).to(dev)
    nb = hists[:, :, 255]                                           # [G, ncol] entries with count >= 255
    lab = torch.empty(sum(lens), dtype=torch.int32, device=dev)
    at, big, lab_off = 0, [], []
    for j, (k, c) in enumerate(merged):
        lab_off.append(at)
        big.append(K.label_shard(k, c, diff[j], int(nb[r, j]), lab[at:at + lens[j]]) if lens[j] else None)
        at += lens[j]
    STATS["distributed_orders"] += 1
    tot_big = int(nb.sum())
    if tot_big == 0:
        return lab, pre
    mine = [(b[1] << 32) | (b[0].to(torch.int64) & 0xFFFFFFFF) for b in big if b is not None and b[0].numel()]
    mine = torch.cat(mine) if mine else torch.empty(0, dtype=torch.int64, device=dev)
    everything = _all_gather_v(mine, sizes=[int(v) for v in nb.sum(1).tolist()])
    # rank-major, column-minor segments -> column-major (owners in rank order = key order)
    seg = nb.reshape(-1)
    goff = torch.zeros(G * ncol + 1, dtype=torch.int64)
    goff[1:] = torch.cumsum(seg, 0)
    col_tot = nb.sum(0)
    col_start = torch.zeros(ncol + 1, dtype=torch.int64)
    col_start[1:] = torch.cumsum(col_tot, 0)
    dst = (col_start[:-1].unsqueeze(0) + (torch.cumsum(nb, 0) - nb)).reshape(-1)   # [G, ncol] -> flat
    bkeys, bcnts = K.exchange_unpack(everything, goff.tolist(), dst.tolist(), tot_big)
    del bkeys
    tags = torch.repeat_interleave(torch.arange(ncol, dtype=torch.int64, device=dev), col_tot.to(dev))
    comp = (tags << 31) | ((1 << 31) - 1 - bcnts)    # (column, count descending); stable: key ascending
    order = _stable_order(comp)
    pos = torch.arange(tot_big, dtype=torch.int64, device=dev) - col_start[:-1].to(dev)[tags]
    lab_cm = torch.empty(tot_big, dtype=torch.int32, device=dev)
    lab_cm[order] = pos.to(torch.int32)              # tags are non-decreasing: tags[order] == tags
    # this owner's entries of column j sit at dst[r, j] .. + nb[r, j] of the column-major array
    idx, val = [], []
    for j, b in enumerate(big):
        n_own = int(nb[r, j])
        if b is None or n_own == 0:
            continue
        d0 = int(dst[r * ncol + j])
        idx.append(b[2].to(torch.int64) + lab_off[j])
        val.append(lab_cm[d0:d0 + n_own])
    if idx:
        lab[torch.cat(idx)] = torch.cat(val)
    return lab, pre


def _exchange_rows_hip(xb, rng_h, G, ncol, ordered=False):
    """Send side on the device path: count matrix, send buffer grouped by (owner, column), the
    all-to-all(v).  ordered: every list is key-sorted -- a group is a contiguous slice of its
    column and is copied in key order (the owner merges sorted runs); else the order of the rows
    inside a group is unspecified (cursor atomics) and the owner sorts."""
    big = torch.iinfo(torch.int64).max
    los, widths = [], []
    for j in range(ncol):
        lo, hi = -rng_h[j][0], rng_h[j][1]
        if hi == -big:  # no entry on any rank
            lo, hi = 0, 0
        los.append(lo)
        widths.append(max(1, -(-(hi - lo + 1) // G)))
    send_mat = xb.hist(los, widths, G)
    _mark("group_rows")
    if _backend() == "nccl":
        recv_mat = torch.empty_like(send_mat)
        _count_traffic("all_to_all_single(count matrix)", _nbytes(send_mat), _nbytes(send_mat))
        td.all_to_all_single(recv_mat, send_mat.contiguous())
    else:
        recv_mat = torch.stack([m[rank()] for m in _all_gather_same(send_mat.contiguous())])
    send_h, recv_h = send_mat.cpu(), recv_mat.cpu()
    starts = torch.zeros(G * ncol, dtype=torch.int64)
    starts[1:] = torch.cumsum(send_h.reshape(-1), 0)[:-1]
    if ordered:
        first_row = torch.cumsum(send_mat, 0) - send_mat   # rows of the column in front of the slice
        rows = xb.pack_ordered(los, widths, G, starts.to(send_mat.device).view(G, ncol), first_row.contiguous())
    else:
        rows = xb.scatter(los, widths, G, starts.to(send_mat.device))
    recv = _all_to_all_v(rows, send_h.sum(1).tolist(), recv_h.sum(1).tolist())
    return send_mat, recv_mat, send_h, recv_h, recv


def _exchange_rows_torch(tables, k64s, lens, rng_h, G, ncol, dev, packed):
    """Send side, general path (any key dtype, host tensors of the gloo tests)."""
    # ---- rows grouped by (owner, column) --------------------------------------------
    own_parts, dest_parts = [], []
    for j, k in enumerate(k64s):
        lo, hi = -rng_h[j][0], rng_h[j][1]
        if lens[j]:
            own = _range_owner(k, lo, hi, G)
        else:
            own = torch.empty(0, dtype=torch.int64, device=dev)
        own_parts.append(own)
        dest_parts.append(own * ncol + j)
    owner_all = torch.cat(own_parts)
    dest = torch.cat(dest_parts)
    if packed:
        rows = _pack_kc(torch.cat(k64s), torch.cat([c.to(torch.int64) for _, c, _ in tables]))
    else:
        rows = torch.stack([torch.cat(k64s), torch.cat([c.to(torch.int64) for _, c, _ in tables])], dim=1)
    if dest.is_cuda:
        # stable radix sort of (destination << 32 | row) words on the destination bits
        from . import kernels as K

        words = K.order_rows(int(dest.numel()), dev, gid=dest, ngroups=G * ncol)
        order = words & 0xFFFFFFFF
    else:  # host stand-in of the CPU tests
        order = torch.argsort(dest, stable=True)
    send_mat = torch.bincount(dest, minlength=G * ncol).to(torch.int64).view(G, ncol)
    _mark("group_rows")
    # ---- count matrix: row g of mine goes to rank g ------------------------------------
    if _backend() == "nccl":
        recv_mat = torch.empty_like(send_mat)
        _count_traffic("all_to_all_single(count matrix)", _nbytes(send_mat), _nbytes(send_mat))
        td.all_to_all_single(recv_mat, send_mat.contiguous())
    else:
        recv_mat = torch.stack([m[rank()] for m in _all_gather_same(send_mat.contiguous())])
    send_h, recv_h = send_mat.cpu(), recv_mat.cpu()
    recv = _all_to_all_v(rows[order].contiguous(), send_h.sum(1).tolist(), recv_h.sum(1).tolist())
    return send_mat, recv_mat, send_h, recv_h, recv


def _merge_and_replicate(tables, dtypes, G, ncol, dev, packed, recv, off, sorted_merge, packed_all,
                         packed_len, device_path, own_hist=None, own_labels=None, pre=None):
    """Owner-side merge (when the sorted merge did not already do it), all-gather of the merged
    shards, the per-column lists every rank ends with."""
    results = {}
    if sorted_merge is None:
        parts = []
        for j in range(ncol):
            pieces = [recv[off[src * ncol + j] : off[src * ncol + j + 1]] for src in range(G)]
            part = torch.cat(pieces) if G > 1 else pieces[0]
            if packed:
                parts.append(_unpack_kc(part, dtypes[j]))
            else:
                parts.append((part[:, 0].contiguous().to(dtypes[j]), part[:, 1].contiguous()))
        live = [j for j in range(ncol) if parts[j][0].numel()]
        if _merge_counts_many_fn is not None:
            results = dict(zip(live, _merge_counts_many_fn([parts[j] for j in live])))
        else:
            results = {j: _merge_counts_fn(*parts[j]) for j in live}
    merged = []  # (sorted merge: packed_all already holds the merged rows of all columns)
    for j in range(ncol if sorted_merge is None else 0):
        if j in results:
            mk, mc = _sort_by_key_fn(*results[j])
            merged.append(_pack_kc(mk.to(torch.int64), mc.to(torch.int64)) if packed
                          else torch.stack([mk.to(torch.int64), mc.to(torch.int64)], dim=1))
        else:
            merged.append(torch.empty((0,) if packed else (0, 2), dtype=torch.int64, device=dev))
    _mark("owner_merge")
    # ---- replicate: every rank gets every owner's share, rank (= key range) order -------------
    if sorted_merge is not None:
        mine, mlen = packed_all, torch.tensor(packed_len, dtype=torch.int64, device=dev)
    else:
        mine = torch.cat(merged)
        mlen = torch.tensor([m.shape[0] for m in merged], dtype=torch.int64, device=dev)
    # (pre: the lengths, the summed scalars and the histogram of the union came with the owners'
    # histograms already, _label_own_shards)
    all_len = pre["all_len"] if pre is not None else torch.stack(_all_gather_same(mlen)).cpu()  # [G, ncol]
    everything = _all_gather_v(mine, sizes=all_len.sum(1).tolist())
    labels_everything = None
    if own_labels is not None:   # the vocabulary positions travel with the rows (4 bytes per entry)
        labels_everything = _all_gather_v(own_labels, sizes=all_len.sum(1).tolist())
    _mark("all_gather")
    goff = torch.zeros(G * ncol + 1, dtype=torch.int64)
    goff[1:] = torch.cumsum(all_len.reshape(-1), 0)
    goff = goff.tolist()
    nsc = max(len(sc) for _, _, sc in tables)
    scal = torch.tensor([list(sc) + [0] * (nsc - len(sc)) for _, _, sc in tables],
                        dtype=torch.int64, device=dev)
    hist_all = None
    if pre is not None:
        hist_all = pre["hist"].to(torch.int32).to(dev)
        hh = pre["hist"].tolist()
        scal = [list(a) + [0] * (nsc - pre["nsc"]) + list(b) for a, b in zip(pre["scal"], hh)]
    else:
        if own_hist is not None:
            # the class histogram of a merged list = the sum of the owners' (every key has ONE
            # owner): each rank histograms 1 / G of the union and the sums ride on this all-reduce
            scal = torch.cat([scal, own_hist], dim=1)
        _all_reduce(scal)
        if own_hist is not None:
            hist_all = scal[:, nsc:].to(torch.int32).contiguous()
        scal = scal.cpu().tolist()
    out = []
    unpacked = label_of = None
    if device_path and packed and everything.is_cuda and everything.numel():
        # column-major in ONE launch: segment (r, j) lands behind the shares of the ranks < r of
        # column j, so every column is one contiguous key-ordered list
        from . import kernels as K

        col_tot = [int(t) for t in all_len.sum(0).tolist()]
        col_start = [0]  # every column starts on a 16-byte boundary of both arrays
        for t in col_tot:
            col_start.append(col_start[-1] + (t + 3) // 4 * 4)
        dst = [0] * (G * ncol)
        for j in range(ncol):
            at = col_start[j]
            for r in range(G):
                dst[r * ncol + j] = at
                at += int(all_len[r, j])
        labs_all = None
        if labels_everything is not None:
            keys_all, cnts_all, labs_all = K.exchange_unpack(everything, goff, dst, col_start[-1],
                                                             extra=labels_everything)
        else:
            keys_all, cnts_all = K.exchange_unpack(everything, goff, dst, col_start[-1])
        unpacked = [(keys_all[col_start[j]:col_start[j] + col_tot[j]],
                     cnts_all[col_start[j]:col_start[j] + col_tot[j]]) for j in range(ncol)]
        if labs_all is not None:
            label_of = [labs_all[col_start[j]:col_start[j] + col_tot[j]] for j in range(ncol)]
    for j in range(ncol):
        if unpacked is not None:
            keys, counts = unpacked[j]
        else:
            seg = torch.cat([everything[goff[r * ncol + j] : goff[r * ncol + j + 1]] for r in range(G)])
            if packed:
                keys, counts = _unpack_kc(seg, dtypes[j])
            else:
                keys, counts = seg[:, 0].contiguous().to(dtypes[j]), seg[:, 1].contiguous()
        info = None
        if keys.numel():
            if hist_all is not None:
                info = dict(sorted_by_key=True, cls_hist=hist_all[j], n_big=int(scal[j][nsc + 255]) & 0xFFFFFFFF,
                            merged=True)
                if label_of is not None:
                    info["label_of"] = label_of[j]
            else:
                info = dict(sorted_by_key=True, cls_hist=_class_hist_fn(counts), n_big=None, merged=True)
        out.append((keys, counts, scal[j][: len(tables[j][2])], info))
    # n_big of every column: ONE read-back of the 256th histogram words
    infos = [o[3] for o in out if o[3] is not None and o[3]["n_big"] is None]
    if infos:
        nb = torch.stack([i["cls_hist"][255] for i in infos]).to(torch.int64).cpu().tolist()
        for i, v in zip(infos, nb):
            i["n_big"] = int(v) & 0xFFFFFFFF
    _mark("unpack_hist")
    return out


def merge_groups(comp: Dict, nkeys: int, nvals: int, sumsq=False, minmax=False) -> Dict:
    """Same for multi-key aggregate tables (JoinGroupby / TargetEncoding / combos)."""
    from . import kernels as K

    G = world_size()
    if G == 1:
        return comp
    keys = comp["keys"]
    nm64 = comp["null_mask"].to(torch.int64)
    owner = _owner_fn(list(keys) + [nm64], G)
    cols = list(keys) + [nm64, comp["size"], comp["count"]] + comp["sum"] + comp["sumsq"] + \
        comp["min"] + comp["max"]
    recv = exchange_rows(cols, owner)
    it = iter(recv)
    rkeys = [next(it) for _ in range(nkeys)]
    rnm = next(it).to(torch.uint8)
    rsize, rcount = next(it), next(it)
    rsum = [next(it) for _ in comp["sum"]]
    rsq = [next(it) for _ in comp["sumsq"]]
    rmin = [next(it) for _ in comp["min"]]
    rmax = [next(it) for _ in comp["max"]]
    tab = K.GroupbyTable(nkeys, nvals, max(64, 2 * int(rsize.numel())), sumsq=sumsq, minmax=minmax)
    tab.merge(rkeys, rnm, rsize, rcount, rsum, rsq, rmin, rmax)
    mine = tab.compact()
    flat = (list(mine["keys"]) + [mine["null_mask"], mine["size"], mine["count"]] + list(mine["sum"])
            + list(mine["sumsq"]) + list(mine["min"]) + list(mine["max"]))
    it = iter(gather_rows(flat))  # the whole table in one all-gather(v)
    out = dict(
        keys=[next(it) for _ in mine["keys"]],
        null_mask=next(it),
        size=next(it),
        count=next(it),
        sum=[next(it) for _ in mine["sum"]],
        sumsq=[next(it) for _ in mine["sumsq"]],
        min=[next(it) for _ in mine["min"]],
        max=[next(it) for _ in mine["max"]],
    )
    out["n"] = int(out["size"].numel())
    return out


def merge_string_luts(lut: Optional[dict]) -> Optional[dict]:
    """Union of the per-rank {surrogate -> string} dictionaries (host objects).

    COLLECTIVE: with more than one rank every rank must call it for the same column, also a
    rank whose shard was empty (lut None) -- it contributes an empty dictionary.  Returns None
    only when no rank had strings."""
    if world_size() == 1:
        return lut
    gathered = [None] * world_size()
    td.all_gather_object(gathered, lut)
    if all(d is None for d in gathered):
        return None
    out = {}
    for d in gathered:
        out.update(d or {})
    return out


def barrier():
    if world_size() > 1:
        td.barrier()


def is_first_rank_with(value) -> bool:
    """True on the lowest rank among those that passed an equal ``value`` (host object;
    collective).  Used to elect ONE writer per shared output directory."""
    if world_size() == 1:
        return True
    gathered = [None] * world_size()
    td.all_gather_object(gathered, value)
    return gathered.index(value) == rank()
