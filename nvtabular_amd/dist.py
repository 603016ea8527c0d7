"""Multi-GPU merge of fit statistics: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference moves per-partition groupby frames between dask workers with a
tree reduce over TCP/UCX and a disk round trip (categorify.py:1423-1529).  Here
each rank keeps its partial tables in HBM and the merge is ONE exchange per fit:

  Categorify / JoinGroupby / TargetEncoding fit
      owner(key) = h32(key) % world          (hash-partitioned, like split_out)
      all-to-all(v) of the (key, count[, sums...]) rows to their owners   <- xGMI, all 7 links busy
      owner-side merge (nvt_count_merge_* / nvt_gb_merge)
      all-gather of the merged shards -> every rank holds the full table and runs the
      same deterministic finalisation (sort, thresholds), so vocabularies are identical
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


def world_size() -> int:
    return td.get_world_size() if td.is_available() and td.is_initialized() else 1


def rank() -> int:
    return td.get_rank() if td.is_available() and td.is_initialized() else 0


def _backend() -> str:
    return td.get_backend() if world_size() > 1 else "none"


# --------------------------------------------------------------------------
# collectives with a gloo-safe fallback
# --------------------------------------------------------------------------
def all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    if world_size() > 1:
        td.all_reduce(t, op=td.ReduceOp.SUM)
    return t


def _nan_reduce(t: torch.Tensor, op) -> torch.Tensor:
    if world_size() == 1:
        return t
    big = float("inf") if op == td.ReduceOp.MIN else float("-inf")
    x = torch.where(torch.isnan(t), torch.full_like(t, big), t)
    td.all_reduce(x, op=op)
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
        td.all_to_all_single(recv, send_counts)
    else:
        gathered = [torch.empty_like(send_counts) for _ in range(G)]
        td.all_gather(gathered, send_counts)
        r = rank()
        recv = torch.stack([g[r] for g in gathered])
    return recv


def _all_to_all_v(send: torch.Tensor, send_counts: List[int], recv_counts: List[int]) -> torch.Tensor:
    """Variable-size all-to-all of a 1-D tensor already grouped by destination rank."""
    G = world_size()
    out = torch.empty(sum(recv_counts), dtype=send.dtype, device=send.device)
    if _backend() == "nccl":
        td.all_to_all_single(out, send, output_split_sizes=recv_counts, input_split_sizes=send_counts)
        return out
    # gloo: pairwise exchange (CPU tests only)
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
        buf = torch.empty(recv_counts[peer], dtype=send.dtype, device=send.device)
        td.recv(buf, peer)
        out[r_off[peer] : r_off[peer + 1]] = buf
    for q in reqs:
        q.wait()
    return out


def _all_gather_v(t: torch.Tensor) -> torch.Tensor:
    """Concatenate every rank's 1-D tensor (variable length), rank order."""
    G = world_size()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.empty_like(n) for _ in range(G)]
    td.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes) if sizes else 0
    pad = torch.zeros(m, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    bufs = [torch.empty_like(pad) for _ in range(G)]
    td.all_gather(bufs, pad)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])


def exchange_rows(columns: Sequence[torch.Tensor], owner: torch.Tensor) -> List[torch.Tensor]:
    """Send row i of every column to rank owner[i]; returns the received columns."""
    G = world_size()
    order = torch.argsort(owner, stable=True)
    send_counts_t = torch.bincount(owner.to(torch.int64), minlength=G).to(torch.int64)
    recv_counts_t = _all_to_all_counts(send_counts_t)
    sc, rc = send_counts_t.cpu().tolist(), recv_counts_t.cpu().tolist()
    return [_all_to_all_v(c[order].contiguous(), sc, rc) for c in columns]


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


_owner_fn: Callable = _hip_owner
_merge_counts_fn: Callable = _hip_merge_counts


def set_backend_fns(owner_fn=None, merge_counts_fn=None):
    """Test hook: replace the HIP owner-hash / owner-merge steps (gloo CPU tests)."""
    global _owner_fn, _merge_counts_fn
    _owner_fn = owner_fn or _hip_owner
    _merge_counts_fn = merge_counts_fn or _hip_merge_counts


# --------------------------------------------------------------------------
# fit-statistics merges
# --------------------------------------------------------------------------
def merge_counts(keys: torch.Tensor, counts: torch.Tensor, nulls: int):
    """Global (key -> count) table from per-rank tables; identical on every rank."""
    G = world_size()
    if G == 1:
        return keys, counts, nulls
    owner = _owner_fn([keys], G)
    rk, rc = exchange_rows([keys, counts], owner)
    mk, mc = _merge_counts_fn(rk, rc)
    keys = _all_gather_v(mk.contiguous())
    counts = _all_gather_v(mc.contiguous())
    n = torch.tensor([nulls], dtype=torch.int64, device=keys.device)
    td.all_reduce(n)
    return keys, counts, int(n.item())


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
    out = dict(
        keys=[_all_gather_v(k) for k in mine["keys"]],
        null_mask=_all_gather_v(mine["null_mask"]),
        size=_all_gather_v(mine["size"]),
        count=_all_gather_v(mine["count"]),
        sum=[_all_gather_v(t) for t in mine["sum"]],
        sumsq=[_all_gather_v(t) for t in mine["sumsq"]],
        min=[_all_gather_v(t) for t in mine["min"]],
        max=[_all_gather_v(t) for t in mine["max"]],
    )
    out["n"] = int(out["size"].numel())
    return out


def merge_string_luts(lut: Optional[dict]) -> Optional[dict]:
    """Union of the per-rank {surrogate -> string} dictionaries (host objects)."""
    if world_size() == 1 or lut is None:
        return lut
    gathered = [None] * world_size()
    td.all_gather_object(gathered, lut)
    out = {}
    for d in gathered:
        out.update(d or {})
    return out
