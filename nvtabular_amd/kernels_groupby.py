"""Groupby kernels: the hash-table groupby (nvt_gb_*), the sort-path groupby of one int32 key column
(nvt_sgb_*), row ordering / segmented aggregation (ops/groupby.py) and the column-wise TargetEncoding
apply kernels.

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


# --------------------------------------------------------------------------
# multi-key groupby tables
# --------------------------------------------------------------------------
class GroupbyTable:
    """nkeys-column groupby-aggregate table (JoinGroupby / TargetEncoding / combo)."""

    def __init__(self, nkeys: int, nvals: int, capacity: int, sumsq=False, minmax=False):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.nkeys, self.nvals = nkeys, nvals
        self.flags = (_lib.NVT_GB_SUMSQ if sumsq else 0) | (_lib.NVT_GB_MINMAX if minmax else 0)
        self.capacity = K.next_pow2(capacity)
        self.device = torch.device("cuda", torch.cuda.current_device())
        # arrays in a torch block (caching allocator: no hipMalloc / hipFree, which synchronise
        # the device, on the fit path); the C handle only carves it up
        nbytes = C.c_uint64()
        check(self.lib.nvt_gb_table_bytes(nkeys, nvals, self.flags, self.capacity, C.byref(nbytes)),
              "nvt_gb_table_bytes")
        self._mem = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
        self._ws = None
        h = C.c_void_p()
        check(self.lib.nvt_gb_create_in(nkeys, nvals, self.flags, self.capacity,
                                        self._mem.data_ptr(), nbytes.value, C.byref(h)),
              "nvt_gb_create_in")
        self.handle = h
        self.clear()

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                self.lib.nvt_gb_destroy(h)
            except Exception:
                pass
            self.handle = None

    def clear(self):
        check(self.lib.nvt_gb_clear(self.handle, K.stream_ptr()), "nvt_gb_clear")

    def state(self) -> List[int]:
        # mailbox read-back of the device state words (no blocking runtime wait)
        return K.read_back_ptr(self.lib.nvt_gb_state_ptr(self.handle), _lib.STATE_WORDS,
                             self.device.index)

    def update(self, keys, key_valid, vals, val_valid):
        keys = [K.widen_i64(k) for k in keys]
        vals = [K.aligned(v.view(torch.uint8) if v.dtype == torch.bool else v) for v in vals]
        n = keys[0].numel()
        need = C.c_uint64()
        check(self.lib.nvt_gb_update_ws_bytes(n, C.byref(need)), "nvt_gb_update_ws_bytes")
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
            check(self.lib.nvt_gb_set_workspace(self.handle, self._ws.data_ptr(), need.value),
                  "nvt_gb_set_workspace")
        kp = _lib.ptr_array([k.data_ptr() for k in keys])
        kv = _lib.ptr_array([K.ptr(v) for v in key_valid])
        vp = _lib.ptr_array([v.data_ptr() for v in vals])
        vv = _lib.ptr_array([K.ptr(v) for v in val_valid])
        vd = (C.c_int * max(1, len(vals)))(*[K.dtype_code(v.dtype) for v in vals])
        check(self.lib.nvt_gb_update(self.handle, kp, kv, vp, vd, vv, n, K.stream_ptr()),
              "nvt_gb_update")

    def merge(self, keys, null_mask, size, count, sums, sumsqs, mins, maxs):
        n = keys[0].numel()
        f = lambda lst: _lib.ptr_array([K.ptr(t) for t in lst]) if lst else None  # noqa: E731
        check(
            self.lib.nvt_gb_merge(
                self.handle, _lib.ptr_array([k.data_ptr() for k in keys]), K.ptr(null_mask),
                K.ptr(size), K.ptr(count), f(sums), f(sumsqs), f(mins), f(maxs), n, K.stream_ptr(),
            ),
            "nvt_gb_merge",
        )

    def compact(self):
        """dict(keys=[...], null_mask, size, count, sum=[...], sumsq=[...], min=[...], max=[...])"""
        st = self.state()
        g = st[_lib.ST_OCCUPIED]
        dev = self.device
        keys = [torch.empty(g, dtype=torch.int64, device=dev) for _ in range(self.nkeys)]
        nm = torch.empty(g, dtype=torch.uint8, device=dev)
        size = torch.empty(g, dtype=torch.int64, device=dev)
        count = torch.empty(g, dtype=torch.int64, device=dev)
        mk = lambda on: (  # noqa: E731
            [torch.empty(g, dtype=torch.float64, device=dev) for _ in range(self.nvals)] if on else []
        )
        sums = mk(True)
        sumsqs = mk(self.flags & _lib.NVT_GB_SUMSQ)
        mins = mk(self.flags & _lib.NVT_GB_MINMAX)
        maxs = mk(self.flags & _lib.NVT_GB_MINMAX)
        out_n = torch.zeros(1, dtype=torch.int64, device=dev)
        f = lambda lst: _lib.ptr_array([t.data_ptr() for t in lst]) if lst else None  # noqa: E731
        check(
            self.lib.nvt_gb_compact(
                self.handle, f(keys), nm.data_ptr(), size.data_ptr(), count.data_ptr(), f(sums),
                f(sumsqs), f(mins), f(maxs), out_n.data_ptr(), K.stream_ptr(),
            ),
            "nvt_gb_compact",
        )
        # index_table: nvt_gb_compact also stored every group's position in its slot, so this
        # table answers lookups for exactly these groups (no nvt_gb_index_build of a second one)
        return dict(keys=keys, null_mask=nm, size=size, count=count, sum=sums, sumsq=sumsqs,
                    min=mins, max=maxs, n=g, index_table=self)

    def index_build(self, keys, null_mask):
        n = keys[0].numel() if keys else 0
        check(
            self.lib.nvt_gb_index_build(
                self.handle, _lib.ptr_array([k.data_ptr() for k in keys]), K.ptr(null_mask), n,
                K.stream_ptr(),
            ),
            "nvt_gb_index_build",
        )

    def lookup(self, keys, key_valid) -> torch.Tensor:
        keys = [K.widen_i64(k) for k in keys]
        n = keys[0].numel()
        out = torch.empty(n, dtype=torch.int64, device=keys[0].device)
        check(
            self.lib.nvt_gb_lookup(
                self.handle, _lib.ptr_array([k.data_ptr() for k in keys]),
                _lib.ptr_array([K.ptr(v) for v in key_valid]), n, out.data_ptr(), K.stream_ptr(),
            ),
            "nvt_gb_lookup",
        )
        return out


def sorted_groupby_eligible(keys: torch.Tensor, key_valid, n: int, kfold: int = 1) -> bool:
    """One int32 / int64 key column without a validity bitmap, enough rows, row index + fold in
    32 bits.  (int64 columns additionally need keys spanning less than 2^32: sorted_groupby
    finds out and returns None otherwise.)"""
    if not K.SORTED_GROUPBY or keys.dtype not in (torch.int32, torch.int64) or key_valid is not None:
        return False
    fb = (kfold - 1).bit_length()
    return (K.SORTED_GROUPBY_MIN_ROWS <= n < (1 << 30) and n <= (1 << (32 - fb))
            and 1 <= kfold <= K.SORTED_GROUPBY_MAX_KFOLD)


def sorted_groupby(keys: torch.Tensor, fold: Optional[torch.Tensor], kfold: int, vals, val_valid,
                   sumsq=False, minmax=False, cap_hint: int = 0, te_records=False):
    """nvt_sgb_sort + nvt_sgb_regroup + nvt_sgb_reduce: groups of ONE int32 key column, dense
    and ordered by key.  Returns the dict GroupbyTable.compact() returns (keys as int64, all-zero
    null mask, count == size: no null keys on this path) plus ``keys32``, ``sorted``, ``shared``
    and -- with folds (TargetEncoding) -- ``fold`` = dict(kfold, size[g * kfold],
    sum[j][g * kfold], records); size / sum are then the totals over the folds.
    Inside ``pass_memo`` the sorted words and the group ids of a key column are computed once for
    all aggregates on it (the second one is a single reduction, without a read-back).
    int64 key columns: one more read-back ({min, max} of the column); None when the keys span
    2^32 or more (the caller falls back to the hash tables).  ``key_offset`` is what the flat
    index subtracts from a column value (0 for int32 columns)."""
    _lib.require_gpu()
    lib = _lib.load()
    dev = keys.device
    n = int(keys.numel())
    keys = keys.contiguous()
    kdt = K.dtype_code(keys.dtype)
    vals = [K.aligned(v.view(torch.uint8) if v.dtype == torch.bool else v) for v in vals]
    nvals = len(vals)
    flags = (_lib.NVT_GB_SUMSQ if sumsq else 0) | (_lib.NVT_GB_MINMAX if minmax else 0)
    memo_key = ("sgb", keys.data_ptr(), n, keys._version)
    _memo = K.current_pass_memo()
    hit = _memo.get(memo_key) if _memo is not None else None
    if hit is not None and hit["bias"] is None:
        return None  # (int64 keys too far apart: found out by an earlier aggregate of this pass)
    if hit is None or not (kfold == 1 or (hit["kfold"] == kfold and hit["fold"] == K.ptr(fold))):
        if hit is not None:
            bias = hit["bias"]
        elif keys.dtype == torch.int32:
            bias = -(1 << 31)
        else:
            mm = torch.empty(2, dtype=torch.int64, device=dev)
            check(lib.nvt_key_minmax(keys.data_ptr(), kdt, n, mm.data_ptr(), K.stream_ptr()),
                  "nvt_key_minmax")
            lo, hi = (int(v) for v in K.read_back(mm).tolist())
            bias = lo if hi - lo < (1 << 32) else None
            if bias is None:
                if _memo is not None:
                    # (keys held: the address cannot be recycled for another column in this pass)
                    _memo[memo_key] = dict(bias=None, kfold=0, fold=None, groups=None, keys=keys)
                return None
        need = C.c_uint64()
        check(lib.nvt_sgb_sort_ws_bytes(n, C.byref(need)), "nvt_sgb_sort_ws_bytes")
        sort_ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        sp, rbc = C.c_void_p(), C.c_int()
        check(lib.nvt_sgb_sort(keys.data_ptr(), kdt, bias, K.ptr(fold), kfold, n, sort_ws.data_ptr(),
                               C.byref(sp), C.byref(rbc), K.stream_ptr()), "nvt_sgb_sort")
        # (a re-sort WITH folds after an aggregate without: the groups are the same keys -- their
        # key lists and what hangs off them, the lookup index, are taken over below)
        hit = dict(sorted=sp.value, rb=rbc.value, kfold=kfold, fold=K.ptr(fold), ws=sort_ws,
                   keys=keys, fold_t=fold, groups=None, bias=bias,
                   prev_groups=hit["groups"] if hit is not None else None)
        if _memo is not None:
            _memo[memo_key] = hit
    def regroup(cap):
        # group ids: words regrouped with the kfold of the SORT (an aggregate without folds
        # divides the slots); the group count starts its way to the host behind the launch
        wk = hit["kfold"]
        need = C.c_uint64()
        check(lib.nvt_sgb_regroup_ws_bytes(n, C.byref(need)), "nvt_sgb_regroup_ws_bytes")
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        words = torch.empty(n, dtype=torch.int64, device=dev)
        state = torch.empty(_lib.STATE_WORDS, dtype=torch.int64, device=dev)
        k64 = torch.empty(cap, dtype=torch.int64, device=dev)
        k32 = torch.empty(cap, dtype=torch.int32, device=dev)
        check(lib.nvt_sgb_regroup(hit["sorted"], hit["rb"], wk, hit["bias"], n, cap, k64.data_ptr(),
                                  k32.data_ptr(), words.data_ptr(), state.data_ptr(),
                                  ws.data_ptr(), K.stream_ptr()), "nvt_sgb_regroup")
        # ("shared": what outlives the pass -- the lookup index of these groups, built once)
        return dict(words=words, kfold=wk, k64=k64, k32=k32, g=None, cap=cap, state=state, shared={},
                    pending=K.PendingReadBack(state), ws=ws)

    def reduce(grp):
        # arrays sized by the group count when the host knows it, by the capacity of the regroup
        # launch otherwise (the first aggregate on a key column: the reduction is enqueued BEHIND
        # the read-back of the count, the device works on it while the host waits)
        cap = max(grp["g"] if grp["g"] is not None else grp["cap"], 1)
        slots = cap * kfold
        size = torch.empty(slots, dtype=torch.int64, device=dev)
        mk = lambda on, m=slots: (  # noqa: E731
            torch.empty((nvals, m), dtype=torch.float64, device=dev) if on and nvals else None)
        fsum, fsq, fmin, fmax = mk(True), mk(sumsq), mk(minmax), mk(minmax)
        tsize = torch.empty(cap, dtype=torch.int64, device=dev) if kfold > 1 else None
        tsum = mk(kfold > 1, cap)
        rec = (torch.empty((nvals, cap, 2 * (kfold + 1)), dtype=torch.float64, device=dev)
               if te_records and kfold > 1 and nvals else None)
        vp = _lib.ptr_array([v.data_ptr() for v in vals])
        vv = _lib.ptr_array([K.ptr(v) for v in val_valid])
        vd = (C.c_int * max(1, nvals))(*[K.dtype_code(v.dtype) for v in vals])
        # value columns in the order of the words: the first aggregate of a pass gathers a column
        # by row (one random sector per row) and leaves it behind in sorted order, the next
        # aggregate on the same words (JoinGroupby after TargetEncoding on the same target) reads
        # that copy streaming
        sv = grp.setdefault("sorted_vals", {})
        s_in, s_out = [None] * nvals, [None] * nvals
        for j, v in enumerate(vals):
            if val_valid[j] is not None or v.dtype == torch.int64 or not K.SHARE_SORTED_VALUES:
                continue
            skey = (v.data_ptr(), v.dtype, v._version)
            have = sv.get(skey)
            if have is not None:
                s_in[j] = have[0]
            elif K.current_pass_memo() is not None:
                s_out[j] = torch.empty(n, dtype=v.dtype, device=dev)
                sv[skey] = (s_out[j], v)   # (v held: its address cannot be recycled in this pass)
        check(lib.nvt_sgb_reduce(
            grp["words"].data_ptr(), grp["kfold"], kfold, vp, vd, vv, nvals, flags, n, cap, size.data_ptr(),
            K.ptr(fsum), K.ptr(fsq), K.ptr(fmin), K.ptr(fmax), K.ptr(tsize), K.ptr(tsum), K.ptr(rec),
            grp["state"].data_ptr(), _lib.ptr_array([K.ptr(t) for t in s_in]),
            _lib.ptr_array([K.ptr(t) for t in s_out]), K.stream_ptr()), "nvt_sgb_reduce")
        return size, fsum, fsq, fmin, fmax, tsize, tsum, rec

    grp = hit["groups"]
    if grp is None:
        wk = hit["kfold"]
        cap = min(n, cap_hint + cap_hint // 4 + 1024) if cap_hint > 0 else n
        cap = min(cap, (0xFFFFFFFE // wk) - 1)
        grp = regroup(cap)
    while True:
        size, fsum, fsq, fmin, fmax, tsize, tsum, rec = reduce(grp)
        if grp["g"] is not None:
            break
        st = grp.pop("pending").get().tolist()
        g = int(st[_lib.ST_OCCUPIED])
        if not st[_lib.ST_NEED]:
            grp["g"] = g
            grp["k64"], grp["k32"] = grp["k64"][:g], grp["k32"][:g]
            pg = hit.get("prev_groups")
            if pg is not None and pg["g"] == g:
                # same key column, same rows: the same ascending key list.  ONE list (and one
                # lookup index, one merge across partitions) for every aggregate on the column,
                # whatever the order of the operators
                grp["k64"], grp["k32"], grp["shared"] = pg["k64"], pg["k32"], pg["shared"]
            hit["groups"] = grp
            break
        # more groups than the hint allowed: both launches again with the exact count
        K.stat_add("count_relaunches")
        if g * grp["kfold"] >= 0xFFFFFFFE:
            raise _lib.NvtHipError("sorted_groupby: groups * kfold does not fit 32 bits")
        grp = regroup(g)
    g, wk = grp["g"], grp["kfold"]
    nan = float("nan")

    def rows(mat, m):
        return [mat[j, :m] for j in range(nvals)] if mat is not None else []

    def untouched(cols, init):  # a group without a valid value keeps the initial +-inf: NaN, as
        return [torch.where(c == init, torch.full_like(c, nan), c) for c in cols]  # nvt_gb_compact

    out = dict(keys=[grp["k64"]], keys32=grp["k32"],
               null_mask=torch.zeros(g, dtype=torch.uint8, device=dev),
               sumsq=rows(fsq, g), min=untouched(rows(fmin, g), float("inf")),
               max=untouched(rows(fmax, g), float("-inf")), n=g,
               sorted=True, shared=grp["shared"], key_offset=hit["bias"] + (1 << 31))
    if kfold > 1:
        out["size"], out["sum"] = tsize[:g], rows(tsum, g)
        out["fold"] = dict(kfold=kfold, size=size[:g * kfold], sum=rows(fsum, g * kfold),
                           records=[rec[j, :g] for j in range(nvals)] if rec is not None else None)
    else:
        out["size"], out["sum"] = size[:g], rows(fsum, g)
    out["count"] = out["size"]
    return out


def flat_index_for(comp) -> "FlatIndex":
    """The FlatIndex of a sorted_groupby result; aggregates that share their group ids (one key
    column, one pass) share the index too."""
    shared = comp.get("shared")
    if shared is not None and shared.get("index") is not None:
        return shared["index"]
    index = K.FlatIndex(comp["keys32"], comp.get("key_offset", 0))
    if shared is not None:
        shared["index"] = index
    return index


def te_apply_folds(group_all, fold, kfold, sum_all, cnt_all, sum_fold, cnt_fold, p_smooth, y_mean,
                   out_dtype=torch.float32):
    _lib.require_gpu()
    n = group_all.numel()
    out = torch.empty(n, dtype=out_dtype, device=group_all.device)
    check(
        _lib.load().nvt_te_apply_folds(
            group_all.data_ptr(), fold.contiguous().data_ptr(), int(kfold), sum_all.data_ptr(),
            cnt_all.data_ptr(), sum_fold.data_ptr(), cnt_fold.data_ptr(), n, float(p_smooth),
            float(y_mean), out.data_ptr(), K.dtype_code(out_dtype), K.stream_ptr(),
        ),
        "nvt_te_apply_folds",
    )
    return out


def _order_ws(n: int, device) -> torch.Tensor:
    need = C.c_uint64()
    check(_lib.load().nvt_order_rows_ws_bytes(n, C.byref(need)), "nvt_order_rows_ws_bytes")
    return torch.empty(need.value, dtype=torch.uint8, device=device)


def order_rows(n: int, device, sort_keys=(), gid: Optional[torch.Tensor] = None, ngroups: int = 0):
    """Row order for the Groupby operator: stable by ``sort_keys`` (most significant first; each a
    (column tensor, validity, ascending) triple), then by group id (-1 = null key, last).
    Returns int64 words: low 32 bits = row index, high half = group id (when gid is given)."""
    lib = _lib.load()
    ws = _order_ws(n, device)
    perm = torch.empty(n, dtype=torch.int64, device=device)
    cur = None
    key64 = torch.empty(n, dtype=torch.int64, device=device) if sort_keys else None
    for data, valid, ascending in reversed(list(sort_keys)):  # LSD: least significant key first
        data = data.view(torch.uint8) if data.dtype == torch.bool else data.contiguous()
        check(lib.nvt_sort_key_u64(data.data_ptr(), K.dtype_code(data.dtype), K.ptr(valid), n,
                                   1 if ascending else 0, key64.data_ptr(), K.stream_ptr()),
              "nvt_sort_key_u64")
        check(lib.nvt_order_rows(key64.data_ptr(), None, 0, K.ptr(cur), n, perm.data_ptr(),
                                 ws.data_ptr(), K.stream_ptr()), "nvt_order_rows")
        cur = perm
    if gid is not None:
        gid = gid.contiguous()
        check(lib.nvt_order_rows(None, gid.data_ptr(), int(ngroups), K.ptr(cur), n, perm.data_ptr(),
                                 ws.data_ptr(), K.stream_ptr()), "nvt_order_rows")
    elif cur is None:
        perm = torch.arange(n, dtype=torch.int64, device=device)
    return perm


def seg_aggregate(words: torch.Tensor, ngroups: int, vals, val_valid, sumsq=False, minmax=False):
    """(size int64[G], count int64[V, G], sum, sumsq or None, min or None, max or None) over rows
    ordered by group (``words`` from order_rows with gid): ONE segmented-reduction launch."""
    lib = _lib.load()
    dev = words.device
    nv = len(vals)
    vals = [K.aligned(v.view(torch.uint8) if v.dtype == torch.bool else v) for v in vals]
    size = torch.zeros(ngroups, dtype=torch.int64, device=dev)
    count = torch.zeros(max(nv, 1), ngroups, dtype=torch.int64, device=dev)
    sm = torch.zeros(max(nv, 1), ngroups, dtype=torch.float64, device=dev)
    sq = torch.zeros(nv, ngroups, dtype=torch.float64, device=dev) if (sumsq and nv) else None
    mn = torch.full((nv, ngroups), float("inf"), dtype=torch.float64, device=dev) if (minmax and nv) else None
    mx = torch.full((nv, ngroups), float("-inf"), dtype=torch.float64, device=dev) if (minmax and nv) else None
    vd = (C.c_int * max(1, nv))(*[K.dtype_code(v.dtype) for v in vals])
    check(lib.nvt_seg_aggregate(words.data_ptr(), words.numel(), int(ngroups),
                                _lib.ptr_array([v.data_ptr() for v in vals]), vd,
                                _lib.ptr_array([K.ptr(v) for v in val_valid]), nv, size.data_ptr(),
                                count.data_ptr(), sm.data_ptr(), K.ptr(sq), K.ptr(mn), K.ptr(mx),
                                K.stream_ptr()), "nvt_seg_aggregate")
    return size, count, sm, sq, mn, mx


def gather(src: torch.Tensor, group: torch.Tensor, miss: float, out_dtype: torch.dtype):
    _lib.require_gpu()
    src = src.to(torch.float64).contiguous()
    out = torch.empty(group.numel(), dtype=out_dtype, device=group.device)
    check(
        _lib.load().nvt_gather_f64(src.data_ptr(), group.data_ptr(), group.numel(), float(miss),
                                   out.data_ptr(), K.dtype_code(out_dtype), K.stream_ptr()),
        "nvt_gather_f64",
    )
    return out


def te_apply(group_all, group_fold, sum_all, cnt_all, sum_fold, cnt_fold, p_smooth, y_mean,
             out_dtype=torch.float32):
    _lib.require_gpu()
    n = group_all.numel()
    out = torch.empty(n, dtype=out_dtype, device=group_all.device)
    check(
        _lib.load().nvt_te_apply(
            group_all.data_ptr(), K.ptr(group_fold), sum_all.data_ptr(), cnt_all.data_ptr(),
            K.ptr(sum_fold), K.ptr(cnt_fold), n, float(p_smooth), float(y_mean), out.data_ptr(),
            K.dtype_code(out_dtype), K.stream_ptr(),
        ),
        "nvt_te_apply",
    )
    return out
