"""Shared driver for multi-key groupby-aggregate fits (JoinGroupby, TargetEncoding).

Mirrors the reference's tree (categorify.py:1344-1540): a fresh table per
partition (_top_level_groupby), folded into the accumulated table
(_mid_level_groupby), finalised once (_bottom_level_groupby) -- with the tables
living in HBM and every level being an ``nvt_gb_*`` kernel.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import pandas as pd
import torch

from .. import kernels as K
from ..device import DeviceFrame, key_view


class GroupAgg:
    def __init__(self, name: str, key_cols: List[str], val_cols: List[str], sumsq=False,
                 minmax=False, hint: int = 0, fold=None, fold_name: str = "", fold_hint: int = 0):
        self.name = name
        self.key_cols = list(key_cols)
        self.val_cols = list(val_cols)
        self.sumsq, self.minmax = sumsq, minmax
        self.table = None
        # one int32 key column: the groups of the first partition come from the sort path
        # (K.sorted_groupby), dense and ordered by key; a second partition demotes them to a
        # hash table and the fit continues as before
        self.sorted_comp = None
        # fold = (fold column, kfold): TargetEncoding's second aggregate on [fold] + key_cols
        # rides on this one -- dense per-(group, fold) statistics on the sort path, a classic
        # GroupAgg (fold_agg) otherwise
        self.fold = fold
        self.fold_name = fold_name
        self.fold_hint = int(fold_hint)
        self.fold_agg = None
        # expected groups per partition: carried from the operator's previous fit (hints dict);
        # unknown -> assume half the rows are distinct (high-cardinality keys are what these
        # operators are used on, and an undersized table costs a full extra pass per retry)
        self.hint = int(hint)
        self.part_hint = 0  # groups of ONE partition (sizes the sort path's per-partition arrays)
        self.strings: Dict[str, dict] = {}
        self.key_dtypes: Dict[str, object] = {}
        self.val_dtypes: Dict[str, object] = {}  # source dtype of every aggregated column

    def _inputs(self, frame: DeviceFrame):
        keys, kvalid = [], []
        for c in self.key_cols:
            col = frame[c]
            if col.fill is not None:
                col = col.materialize()
            k, v = key_view(col)
            keys.append(k)
            kvalid.append(v)
            self.key_dtypes.setdefault(c, "str" if col.strings is not None else col.data.dtype)
            if col.strings is not None:
                self.strings.setdefault(c, {}).update(col.strings)
        vals, vvalid = [], []
        for c in self.val_cols:
            col = frame[c].materialize()
            vals.append(col.data)
            vvalid.append(col.valid)
            self.val_dtypes.setdefault(c, col.data.dtype)
        return keys, kvalid, vals, vvalid

    def _fold_classic(self) -> "GroupAgg":
        if self.fold_agg is None:
            self.fold_agg = GroupAgg(self.fold_name, [self.fold[0]] + self.key_cols, self.val_cols,
                                     hint=self.fold_hint)
        return self.fold_agg

    def _demote(self):
        """A second partition arrives: the sorted groups of the first become hash tables."""
        comp, self.sorted_comp = with_null_group(self.sorted_comp), None
        self.table = _table_from_comp(comp, 1, len(self.val_cols), self.sumsq, self.minmax)
        if "fold" in comp:
            self._fold_classic().table = _table_from_comp(fold_sparse(comp), 2, len(self.val_cols),
                                                          False, False)

    def update(self, frame: DeviceFrame):
        from .. import dist

        keys, kvalid, vals, vvalid = self._inputs(frame)
        n = int(keys[0].numel())
        kfold = self.fold[1] if self.fold else 1
        fold_t = None
        if self.fold:
            fcol = frame[self.fold[0]]
            fold_t = fcol.data if fcol.data.dtype == torch.uint8 and fcol.valid is None else None
        nullgrp = None
        sort_ok = (self.table is None and len(keys) == 1
                   and (self.fold_agg is None or self.fold_agg.table is None)
                   and (not self.fold or fold_t is not None))
        s_keys, s_kvalid, s_vals, s_vvalid, s_fold, s_n = keys, kvalid, vals, vvalid, fold_t, n
        if sort_ok and kvalid[0] is not None and NULL_KEY_GROUP and \
                K.sorted_groupby_eligible(keys[0], None, n, kfold):
            # null KEYS: their rows are one group of their own (the reference groups with
            # dropna=False, categorify.py:1013-1018).  The sort path takes the rows WITH a key
            # (compacted: torch plumbing, one pass per column); the null rows are reduced on the
            # side and travel with the result as comp["nullgrp"]
            s_keys, s_kvalid, s_vals, s_vvalid, s_fold, nullgrp, s_n = _split_null_keys(
                keys[0], kvalid[0], vals, vvalid, fold_t, kfold, self.sumsq, self.minmax)
        if sort_ok and K.sorted_groupby_eligible(s_keys[0], s_kvalid[0], s_n, kfold):
            comp = K.sorted_groupby(s_keys[0], s_fold, kfold, s_vals, s_vvalid, sumsq=self.sumsq,
                                    minmax=self.minmax, cap_hint=self.part_hint or self.hint,
                                    te_records=not K.LOOKUP_IMAGES)
            if comp is not None and nullgrp is not None:
                comp["nullgrp"] = nullgrp
            if comp is not None:  # (None: int64 keys spanning 2^32 or more)
                self.part_hint = max(self.part_hint, comp["n"])
                if self.sorted_comp is None:
                    self.hint = max(self.hint, comp["n"])
                    self.sorted_comp = comp
                    return
                # a further partition: both sides are dense groups ORDERED BY KEY -- merged by
                # the merge-path kernel (_mid_level_groupby, categorify.py:1054-1070 under
                # join_groupby.py:140-173 / target_encoding.py:171-214), still ordered by key: the
                # fit keeps the sort path (flat index, fused lookups) however many partitions
                merged = merge_sorted_comps(self.sorted_comp, comp, self.sumsq, self.minmax)
                if merged is not None:
                    self.hint = max(self.hint, merged["n"])
                    self.sorted_comp = merged
                    return
                # (int64 keys whose union spans 2^32 or more: the hash tables take over -- the
                # accumulated groups are demoted ONCE below and this partition's raw rows go
                # through the hash path exactly once)
        if self.sorted_comp is not None:
            self._demote()
        if self.fold:
            self._fold_classic().update(frame)
        hint = self.hint if self.hint > 0 else max(1 << 12, n // 2)
        # load <= 2/3 on the hinted group count (16-byte slot headers: four per sector, so the
        # longer probe runs of a fuller table mostly stay inside one sector, and a 128 MB head
        # array instead of 256 MB stays in the Infinity Cache: cfg4 16.4 -> 15.7 ms per step)
        cap = K.next_pow2(3 * min(max(hint, 32), max(n, 32)) // 2)
        while True:
            part = K.GroupbyTable(len(keys), len(vals), cap, sumsq=self.sumsq, minmax=self.minmax)
            part.update(keys, kvalid, vals, vvalid)
            st = part.state()
            if not st[K._lib.ST_OVERFLOW] and st[K._lib.ST_OCCUPIED] * 4 <= part.capacity * 3:
                break
            cap *= 4
        self.hint = max(self.hint, st[K._lib.ST_OCCUPIED])
        if self.table is None:
            self.table = part
        else:
            from .categorify import _merge_groups

            self.table = _merge_groups(self.table, part.compact())

    def finalize(self):
        """Compacted groups (after the cross-rank merge), ordered by key for determinism."""
        from .. import dist

        if self.sorted_comp is not None:
            null_at = int(self.sorted_comp["n"]) if "nullgrp" in self.sorted_comp else -1
            comp = with_null_group(self.sorted_comp)
            if dist.world_size() > 1:
                if "fold" in comp:
                    # the ranks merge compacted groups: the dense per-(group, fold) statistics
                    # become the [fold, key] groups of the second aggregate, merged on their own
                    self._fold_classic().sorted_comp = fold_sparse(comp)
                    comp = {k: v for k, v in comp.items() if k != "fold"}
                comp = dist.merge_groups(comp, len(self.key_cols), len(self.val_cols),
                                         sumsq=self.sumsq, minmax=self.minmax)
            elif len(self.key_cols) == 1 and "keys32" in comp:
                index = K.flat_index_for(comp)
                if index.ok():  # else: _Stats builds a hashed index from the keys
                    if null_at >= 0 and int(comp["n"]) > null_at:
                        index.set_null_group(null_at)   # rows with a null key look up the last group
                    comp["index_table"] = index
            return comp
        if self.table is None:
            self.table = K.GroupbyTable(len(self.key_cols), len(self.val_cols), 64,
                                        sumsq=self.sumsq, minmax=self.minmax)
        comp = self.table.compact()
        if dist.world_size() > 1:
            comp = dist.merge_groups(comp, len(self.key_cols), len(self.val_cols),
                                     sumsq=self.sumsq, minmax=self.minmax)
            for c in list(self.strings):
                self.strings[c] = dist.merge_string_luts(self.strings[c])
        return comp


NULL_KEY_GROUP = True


def _split_null_keys(key, valid, vals, vvalid, fold_t, kfold, sumsq, minmax):
    """(keys, key validity, values, value validity, folds) of the rows WITH a key, the statistics
    of the rows without one ("nullgrp"), and the number of rows kept."""
    from ..device import pack_bitmap_device

    n = int(key.numel())
    has = K.unpack_bitmap(valid, n)
    inv = ~has
    dev = key.device
    cvals, cvv, nsum, nsq, nmin, nmax = [], [], [], [], [], []
    f64 = torch.float64
    ninf = torch.tensor(float("inf"), dtype=f64, device=dev)
    size = inv.sum().to(torch.int64)
    fold64 = fold_t.to(torch.int64) if fold_t is not None else None
    fsum = []
    for v, vv in zip(vals, vvalid):
        x = v.to(f64)
        ok = inv & ~torch.isnan(x)
        if vv is not None:
            vok = K.unpack_bitmap(vv, n)
            ok = ok & vok
            cvv.append(pack_bitmap_device(vok[has]))
        else:
            cvv.append(None)
        cvals.append(v[has])
        xz = torch.where(ok, x, torch.zeros((), dtype=f64, device=dev))
        nsum.append(xz.sum())
        if sumsq:
            nsq.append((xz * xz).sum())
        if minmax:
            mn = torch.where(ok, x, ninf).min() if n else ninf
            mx = torch.where(ok, x, -ninf).max() if n else -ninf
            nan = torch.tensor(float("nan"), dtype=f64, device=dev)
            nmin.append(torch.where(torch.isinf(mn), nan, mn))   # (no value: NaN, as nvt_gb_compact)
            nmax.append(torch.where(torch.isinf(mx), nan, mx))
        if fold64 is not None and kfold > 1:
            fsum.append(torch.bincount(fold64[ok], weights=x[ok], minlength=kfold).to(f64))
    grp = dict(size=size, sum=nsum, sumsq=nsq, min=nmin, max=nmax)
    if fold64 is not None and kfold > 1:
        grp["fold"] = dict(size=torch.bincount(fold64[inv], minlength=kfold).to(torch.int64), sum=fsum)
    return ([key[has]], [None], cvals, cvv, fold_t[has] if fold_t is not None else None, grp,
            int(n - int(size.item())))


def _merge_null_groups(x, y):
    if x is None or y is None:
        return x if y is None else y

    def mm(a, b, op):
        pick = torch.minimum if op == "min" else torch.maximum
        return torch.where(torch.isnan(a), b, torch.where(torch.isnan(b), a, pick(a, b)))

    out = dict(size=x["size"] + y["size"], sum=[a + b for a, b in zip(x["sum"], y["sum"])],
               sumsq=[a + b for a, b in zip(x["sumsq"], y["sumsq"])],
               min=[mm(a, b, "min") for a, b in zip(x["min"], y["min"])],
               max=[mm(a, b, "max") for a, b in zip(x["max"], y["max"])])
    if "fold" in x and "fold" in y:
        out["fold"] = dict(size=x["fold"]["size"] + y["fold"]["size"],
                           sum=[a + b for a, b in zip(x["fold"]["sum"], y["fold"]["sum"])])
    return out


def with_null_group(comp):
    """The sort-path result with the group of the null-key rows (comp["nullgrp"]) appended as its
    LAST group (null_mask 1, key 0): what the artifacts, the hash tables and the ranks' merge
    consume.  keys32 (the flat index's keys) stays as it is: the index sends null rows to the
    group through FlatIndex.set_null_group."""
    g0 = comp.get("nullgrp")
    if g0 is None or int(g0["size"].item()) == 0:
        return {k: v for k, v in comp.items() if k != "nullgrp"}
    dev = comp["size"].device
    one = lambda t, dt: t.reshape(1).to(dt)  # noqa: E731
    out = {k: v for k, v in comp.items() if k != "nullgrp"}
    out["keys"] = [torch.cat([comp["keys"][0], torch.zeros(1, dtype=torch.int64, device=dev)])]
    out["null_mask"] = torch.cat([comp["null_mask"], torch.ones(1, dtype=torch.uint8, device=dev)])
    out["size"] = torch.cat([comp["size"], one(g0["size"], torch.int64)])
    # "count" is the count of the KEY column (categorify.py:1004-1018: agg_dict[key] = ["count"]),
    # which skips nulls: 0 for the group of the null keys, whatever its size
    out["count"] = torch.cat([comp["size"], torch.zeros(1, dtype=torch.int64, device=dev)])
    for name in ("sum", "sumsq", "min", "max"):
        out[name] = [torch.cat([a, one(b, torch.float64)]) for a, b in zip(comp[name], g0[name])]
    out["n"] = int(comp["n"]) + 1
    if "fold" in comp and "fold" in g0:
        f, gf = comp["fold"], g0["fold"]
        kfold = f["kfold"]
        rec = None
        if f.get("records") is not None:
            rec = []
            for j, r in enumerate(f["records"]):
                row = torch.empty(2 * (kfold + 1), dtype=torch.float64, device=dev)
                # the reference's counts: the per-key table counts the KEY column (0 for the null
                # keys), the [fold, key] table counts its FIRST key column, the fold (= the rows)
                row.zero_()
                row[0] = g0["sum"][j]
                row[2::2], row[3::2] = gf["sum"][j], gf["size"].to(torch.float64)
                rec.append(torch.cat([r, row.reshape(1, -1)]))
        out["fold"] = dict(kfold=kfold, size=torch.cat([f["size"], gf["size"]]),
                           sum=[torch.cat([a, b]) for a, b in zip(f["sum"], gf["sum"])], records=rec)
    return out


def merge_sorted_comps(a, b, sumsq=False, minmax=False):
    """Two sort-path results (K.sorted_groupby: dense groups of ONE int32 key column, ordered by
    key) -> the same structure for the union of their rows: ONE merge-path pass over the keys
    (nvt_merge_sorted_many with source maps) + one gather-combine per statistic
    (nvt_merge_payload: sums / sizes add, min / max combine, per-fold blocks as rows of kfold
    values, the transform's {sum, count} records as rows of 2 (kfold + 1)).  None when the two
    sides carry int64 key images whose union spans 2^32 keys or more.  Inside a pass (K.pass_memo) the key
    merge of a column is shared by every aggregate on it, like the sort itself."""
    if ("fold" in a) != ("fold" in b):
        return None
    memo = K.current_pass_memo()
    off_a, off_b = int(a.get("key_offset", 0)), int(b.get("key_offset", 0))
    if off_a != off_b:
        # int64 key column: the 32-bit image is key - offset and every partition picks its own
        # offset (from its smallest key).  Both sides are re-based to ONE offset when the union
        # of their keys spans less than 2^32 (one read-back of each side's first / last key):
        # the accumulated side keeps its offset when the partition fits there, else the common
        # offset moves down to the union's smallest key; no merge only for a wider union.
        rkey = ("sgb_rebase", a["keys32"].data_ptr(), int(a["n"]), off_a,
                b["keys32"].data_ptr(), int(b["n"]), off_b)
        rb = memo.get(rkey) if memo is not None else None
        if rb is None:
            def ends(c, off):
                if not int(c["n"]):
                    return None
                e = K.read_back(torch.stack([c["keys32"][0], c["keys32"][int(c["n"]) - 1]]).to(torch.int64))
                return int(e[0]) + off, int(e[1]) + off
            ea, eb = ends(a, off_a), ends(b, off_b)
            span = [e for e in (ea, eb) if e is not None]
            lo = min(e[0] for e in span) if span else 0
            hi = max(e[1] for e in span) if span else 0
            new_off = None
            if lo - off_a >= -(1 << 31) and hi - off_a <= (1 << 31) - 1:
                new_off = off_a
            elif hi - lo <= (1 << 32) - 1:
                new_off = lo + (1 << 31)

            def rebase(c, off):
                if new_off is None or off == new_off:
                    return c["keys32"]
                return (c["keys32"].to(torch.int64) + (off - new_off)).to(torch.int32)
            rb = dict(off=new_off, ka=rebase(a, off_a), kb=rebase(b, off_b), keep=(a["keys32"], b["keys32"]))
            if memo is not None:
                memo[rkey] = rb
        if rb["off"] is None:
            return None
        a = dict(a, keys32=rb["ka"], key_offset=rb["off"])
        b = dict(b, keys32=rb["kb"], key_offset=rb["off"])
    mkey = ("sgb_merge", a["keys32"].data_ptr(), int(a["n"]), b["keys32"].data_ptr(), int(b["n"]))
    hit = memo.get(mkey) if memo is not None else None
    if hit is None:
        (k32, size, sa, sb), = K.merge_sorted_pairs(
            [((a["keys32"], a["size"]), (b["keys32"], b["size"]))], want_src=True)
        hit = dict(k32=k32, size=size, sa=sa, sb=sb, shared={},
                   k64=k32.to(torch.int64) + int(a.get("key_offset", 0)), keep=(a["keys32"], b["keys32"]))
        if memo is not None:
            memo[mkey] = hit
    k32, size, sa, sb = hit["k32"], hit["size"], hit["sa"], hit["sb"]
    g = int(k32.numel())

    def comb(x, y, op="add", width=1):
        return K.merge_payload(sa, sb, x.reshape(-1), y.reshape(-1), op, width)

    ng = _merge_null_groups(a.get("nullgrp"), b.get("nullgrp"))
    out = dict(keys=[hit["k64"]], keys32=k32, n=g, sorted=True, shared=hit["shared"],
               key_offset=a.get("key_offset", 0), size=size, count=size,
               null_mask=torch.zeros(g, dtype=torch.uint8, device=k32.device),
               sum=[comb(x, y) for x, y in zip(a["sum"], b["sum"])],
               sumsq=[comb(x, y) for x, y in zip(a["sumsq"], b["sumsq"])] if sumsq else [],
               min=[comb(x, y, "min") for x, y in zip(a["min"], b["min"])] if minmax else [],
               max=[comb(x, y, "max") for x, y in zip(a["max"], b["max"])] if minmax else [])
    if "fold" in a:
        fa, fb = a["fold"], b["fold"]
        kfold = fa["kfold"]
        if fb["kfold"] != kfold:
            return None
        rs = 2 * (kfold + 1)
        rec = None
        if fa.get("records") is not None and fb.get("records") is not None:
            rec = [comb(x, y, "add", rs).view(g, rs) for x, y in zip(fa["records"], fb["records"])]
        out["fold"] = dict(kfold=kfold, size=comb(fa["size"], fb["size"], "add", kfold),
                           sum=[comb(x, y, "add", kfold) for x, y in zip(fa["sum"], fb["sum"])],
                           records=rec)
    if ng is not None:
        out["nullgrp"] = ng
    return out


def _table_from_comp(comp, nkeys, nvals, sumsq, minmax) -> "K.GroupbyTable":
    tab = K.GroupbyTable(nkeys, nvals, max(64, 2 * int(comp["n"]) + 1), sumsq=sumsq, minmax=minmax)
    tab.merge([k.contiguous() for k in comp["keys"]], comp["null_mask"], comp["size"].contiguous(),
              comp["count"].contiguous(), [c.contiguous() for c in comp["sum"]],
              [c.contiguous() for c in comp["sumsq"]], [c.contiguous() for c in comp["min"]],
              [c.contiguous() for c in comp["max"]])
    return tab


def fold_sparse(comp):
    """Dense per-(group, fold) statistics of the sort path -> the compacted [fold, key] groups
    the reference's second groupby produces (only pairs that have rows)."""
    f = comp["fold"]
    kfold = f["kfold"]
    idx = torch.nonzero(f["size"] > 0).squeeze(1)
    size = f["size"][idx]
    nm = comp.get("null_mask")   # (sort-path groups without a null-key group carry none)
    if nm is None:
        nm = torch.zeros(comp["n"], dtype=torch.uint8, device=size.device)
    # (bit 1 = the key column of the [fold, key] tuple: set for the group of the null-key rows)
    return dict(keys=[idx % kfold, comp["keys"][0][idx // kfold]],
                null_mask=(nm[idx // kfold].to(torch.uint8) << 1),
                size=size, count=size, sum=[c[idx] for c in f["sum"]], sumsq=[], min=[], max=[],
                n=int(idx.numel()))


def stats_frame(agg: GroupAgg, comp, stats, name_sep="_", count_name=None) -> pd.DataFrame:
    """Host frame in the reference's cat_stats layout (categorify.py:1079-1137): key
    columns, <name>_count, <name>_<cont>_{sum,mean,min,max,var,std} as requested."""
    keys = [k.cpu().numpy() for k in comp["keys"]]
    nm = comp["null_mask"].cpu().numpy()
    data = {}
    for j, c in enumerate(agg.key_cols):
        col = keys[j]
        isnull = ((nm >> j) & 1).astype(bool)
        lut = agg.strings.get(c)
        if lut is not None:
            col = np.array([lut.get(int(k)) for k in col], dtype=object)
            col[isnull] = None
        else:
            src = agg.key_dtypes.get(c)
            if isinstance(src, torch.dtype) and src in (torch.int32, torch.uint8, torch.bool):
                col = col.astype({torch.int32: np.int32, torch.uint8: np.uint8,
                                  torch.bool: np.uint8}[src])
            if isnull.any():
                col = col.astype(np.float64)
                col[isnull] = np.nan
        data[c] = col
    base = name_sep.join(agg.key_cols)
    count = comp["count"].cpu().numpy()
    derived = derive_stats(comp, stats)
    if "count" in stats:
        data[f"{base}{name_sep}count"] = count
    if "size" in stats:
        data[f"{base}{name_sep}size"] = comp["size"].cpu().numpy()
    for j, cont in enumerate(agg.val_cols):
        for stat in ("sum", "mean", "min", "max", "var", "std"):
            if stat in stats:
                col = derived[(j, stat)].cpu().numpy()
                if stat in ("sum", "min", "max") and agg.val_dtypes.get(cont) == torch.float32:
                    col = col.astype(np.float32)  # pandas' groupby keeps the float32 of the source
                data[f"{base}{name_sep}{cont}{name_sep}{stat}"] = col
    df = pd.DataFrame(data)
    return df


def derive_stats(comp, stats):
    """(value index, stat) -> float64 tensor [groups]  (categorify.py:1087-1131)."""
    out = {}
    n = comp["count"].to(torch.float64)
    for j in range(len(comp["sum"])):
        x = comp["sum"][j]
        out[(j, "sum")] = x
        if "mean" in stats:
            out[(j, "mean")] = x / n
        if "min" in stats:
            out[(j, "min")] = comp["min"][j]
        if "max" in stats:
            out[(j, "max")] = comp["max"][j]
        if "var" in stats or "std" in stats:
            x2 = comp["sumsq"][j]
            res = x2 - x * x / n
            div = torch.clamp(n - 1, min=1)
            res = res / div
            res = torch.where((n - 1) == 0, torch.full_like(res, float("nan")), res)
            out[(j, "var")] = res
            out[(j, "std")] = torch.sqrt(res)
    return out
