"""Groupby (reference: nvtabular/ops/groupby.py:27-330): collapse the rows of each key group
into one row -- list / first / last aggregations in ``sort_cols`` order plus the conventional
count / sum / mean / min / max / std / var.

Row -> group assignment runs on the multi-key hash tables of the JoinGroupby path
(``nvt_gb_update`` / ``nvt_gb_index_build`` / ``nvt_gb_lookup``); ordering the rows inside the
groups and the segmented reductions are torch device ops -- this operator is graph plumbing
around the hot path, not one of its kernels.  Like the reference
it works partition by partition: shuffle the dataset by the keys first.
"""
from __future__ import annotations

import numpy
import torch

from .. import kernels as K
from ..device import DeviceColumn, DeviceFrame, as_device_frame, key_view, pack_bitmap_device
from ..schema import Tags
from ..selector import ColumnSelector
from ._groupby import GroupAgg
from .base import Operator

_FLOAT32_AGGS = ("mean", "median", "std", "var", "sum")
_INT32_AGGS = ("count", "nunique")
_SUPPORTED = ("list", "first", "last", "count", "sum", "mean", "min", "max", "std", "var")


def is_list_agg(agg, custom=False):
    if custom:
        return agg in ("first", "last")
    return agg in ("list", list, "first", "last")


class Groupby(Operator):
    def __init__(self, groupby_cols=None, sort_cols=None, aggs="list", name_sep="_", ascending=True):
        self.groupby_cols = [groupby_cols] if isinstance(groupby_cols, str) else groupby_cols
        self.sort_cols = [sort_cols] if isinstance(sort_cols, str) else (sort_cols or [])
        self.ascending = ascending
        # groupby.py:88-107: split into list-based and conventional aggregations
        self.list_aggs, self.conv_aggs = {}, {}
        if isinstance(aggs, str):
            aggs = {"__all__": [aggs]}
        elif isinstance(aggs, list):
            aggs = {"__all__": aggs}
        for col, v in aggs.items():
            _aggs = v if isinstance(v, list) else [v]
            conv, lst = [], []
            for a in _aggs:
                if is_list_agg(a):
                    a = "list" if a == list else a
                    if a not in lst:
                        lst.append(a)
                elif a not in conv:
                    conv.append(a)
            for a in conv + lst:
                if a not in _SUPPORTED:
                    raise NotImplementedError(f"Groupby aggregation '{a}' is not supported")
            if conv:
                self.conv_aggs[col] = conv
            if lst:
                self.list_aggs[col] = lst
        self.name_sep = name_sep
        super().__init__()

    # ---- naming (groupby.py:160-186, 262-330) ----------------------------------------------
    def _agg_dicts(self, col_selector):
        names = col_selector.names if isinstance(col_selector, ColumnSelector) else list(col_selector)
        allowed = [c for c in names if c not in self.groupby_cols]

        def ensure(d):
            if "__all__" in d:
                return {c: d["__all__"] for c in allowed}
            return {k: v for k, v in d.items() if k in allowed}

        return ensure(self.list_aggs), ensure(self.conv_aggs)

    def column_mapping(self, col_selector):
        mapping = {}
        for g in self.groupby_cols:
            if g in col_selector.names:
                mapping[g] = [g]
        lst, conv = self._agg_dicts(col_selector)
        for aggs in (lst, conv):
            for col, names in aggs.items():
                for a in names:
                    mapping[self.name_sep.join([col, a])] = [col]
        return mapping

    @property
    def dependencies(self):
        return self.groupby_cols

    def _find_agg(self, col_schema, input_schema):
        mapping = self.column_mapping(ColumnSelector(input_schema.column_names))
        src = mapping[col_schema.name][0]
        return col_schema.name.replace(src, "").lstrip(self.name_sep)

    def _compute_dtype(self, col_schema, input_schema):
        col_schema = super()._compute_dtype(col_schema, input_schema)
        agg = self._find_agg(col_schema, input_schema)
        if agg in _INT32_AGGS:
            return col_schema.with_dtype(numpy.int32)
        if agg in _FLOAT32_AGGS:
            return col_schema.with_dtype(numpy.float32)
        return col_schema

    def _compute_shape(self, col_schema, input_schema):
        agg = self._find_agg(col_schema, input_schema)
        if agg == "list":
            return col_schema.with_shape(is_list=True, is_ragged=True)
        return col_schema.with_shape(is_list=False, is_ragged=False)

    def _compute_tags(self, col_schema, input_schema):
        col_schema = super()._compute_tags(col_schema, input_schema)
        agg = self._find_agg(col_schema, input_schema)
        return col_schema.with_tags([Tags.LIST]) if agg == "list" else col_schema

    # ---- transform ---------------------------------------------------------------------------
    def transform(self, col_selector: ColumnSelector, df):
        frame, was_pandas = as_device_frame(df)
        out = self._transform_frame(col_selector, frame)
        return out.to_pandas() if was_pandas else out

    def _transform_frame(self, col_selector, frame: DeviceFrame) -> DeviceFrame:
        n = len(frame)
        lst, conv = self._agg_dicts(col_selector)
        dev = next(iter(frame.items()))[1].data.device
        # 1. the groups: distinct key tuples (HIP multi-key table), null keys dropped like
        #    pandas' groupby(dropna=True), ordered by key like groupby(sort=True)
        agg = GroupAgg("groupby", self.groupby_cols, [])
        agg.update(frame)
        comp = agg.table.compact()
        keep = comp["null_mask"] == 0
        gkeys = [k[keep] for k in comp["keys"]]
        order = _lexsort(gkeys, agg, self.groupby_cols)
        gkeys = [k[order] for k in gkeys]
        ngroups = int(gkeys[0].numel()) if gkeys else 0
        index = K.GroupbyTable(len(gkeys), 0, max(64, 2 * ngroups + 1))
        index.index_build([k.contiguous() for k in gkeys],
                          torch.zeros(ngroups, dtype=torch.uint8, device=dev))
        keys, valids = [], []
        for c in self.groupby_cols:
            k, v = key_view(frame[c].materialize())
            keys.append(k)
            valids.append(v)
        gid = index.lookup(keys, valids)  # row -> rank of its group, -1 for a null key
        # 2. row order: sort_cols (stable, nulls last), then stable by group -- the stable radix
        #    sort of nvt_sort.hip over (key32 << 32 | row) words (nvt_order_rows); rows of null
        #    keys sort behind the last group and are cut off
        sort_keys = []
        for c in self.sort_cols:
            col = frame[c].materialize()
            sort_keys.append((col.data, col.valid, bool(self.ascending)))
        words = K.order_rows(n, dev, sort_keys, gid, ngroups)
        conts = [c for c in dict.fromkeys(list(lst) + list(conv))]
        cols = {c: frame[c].materialize() for c in conts}
        for c, col in cols.items():
            if col.is_list:
                raise NotImplementedError("Groupby over list columns")
        want = {a for c in conv for a in conv[c]}
        # 3. every conventional aggregate from ONE segmented reduction over the ordered rows
        size, count, sm, sq, mn, mx = K.seg_aggregate(
            words, ngroups, [cols[c].data for c in conts], [cols[c].valid for c in conts],
            sumsq=bool(want & {"std", "var"}), minmax=bool(want & {"min", "max"}))
        offsets = torch.zeros(ngroups + 1, dtype=torch.int64, device=dev)
        torch.cumsum(size, 0, out=offsets[1:])
        n_valid = int(offsets[-1].item())
        perm = (words[:n_valid] & 0xFFFFFFFF)
        # 4. outputs
        out = DeviceFrame()
        for j, c in enumerate(self.groupby_cols):
            if c in col_selector.names:
                src = frame[c]
                data = gkeys[j].to(src.data.dtype) if src.strings is None else gkeys[j]
                out[c] = DeviceColumn(data, None, None, None, src.strings)
        for ci, c in enumerate(conts):
            col = cols[c]
            vals = ok = None
            for a in lst.get(c, []):
                if vals is None:
                    vals = col.data[perm]
                    ok = _valid_bool(col, n)[perm]
                name = self.name_sep.join([c, a])
                if a == "list":
                    vb = None if bool(ok.all()) else pack_bitmap_device(ok)
                    out[name] = DeviceColumn(vals.contiguous(), vb, offsets, None, col.strings)
                else:
                    take_first = (a == "first") == bool(self.ascending)  # groupby.py:287-296
                    pos = offsets[:-1] if take_first else offsets[1:] - 1
                    sel_ok = ok[pos]
                    vb = None if bool(sel_ok.all()) else pack_bitmap_device(sel_ok)
                    out[name] = DeviceColumn(vals[pos].contiguous(), vb, None, None, col.strings)
            for a in conv.get(c, []):
                out[self.name_sep.join([c, a])] = _finish_agg(
                    a, col.data.dtype, count[ci].to(torch.float64), sm[ci],
                    sq[ci] if sq is not None else None, mn[ci] if mn is not None else None,
                    mx[ci] if mx is not None else None)
        return out

    @property
    def output_tags(self):
        return []


def _valid_bool(col: DeviceColumn, n: int) -> torch.Tensor:
    dev = col.data.device
    if col.valid is None:
        ok = torch.ones(n, dtype=torch.bool, device=dev)
    else:
        idx = torch.arange(n, device=dev)
        ok = ((col.valid[idx >> 3] >> (idx & 7).to(torch.uint8)) & 1).to(torch.bool)
    if col.data.dtype.is_floating_point:
        ok = ok & ~torch.isnan(col.data)
    return ok


def _sortable(col: DeviceColumn, ascending: bool) -> torch.Tensor:
    """Sort key with nulls placed last for either direction (pandas na_position='last')."""
    ok = _valid_bool(col, int(col.data.numel()))
    if col.data.dtype.is_floating_point:
        v = col.data.to(torch.float64)
        return torch.where(ok, v, torch.full_like(v, float("inf") if ascending else float("-inf")))
    v = col.data.to(torch.int64)
    lim = torch.iinfo(torch.int64)
    return torch.where(ok, v, torch.full_like(v, lim.max if ascending else lim.min))


def _lexsort(gkeys, agg: GroupAgg, cols) -> torch.Tensor:
    """Ascending lexicographic order of the group key tuples (strings by value, on the host)."""
    if not gkeys or gkeys[0].numel() == 0:
        return torch.zeros(0, dtype=torch.int64, device=gkeys[0].device if gkeys else None)
    if any(c in agg.strings for c in cols):
        import numpy as np

        host = []
        for k, c in zip(gkeys, cols):
            h = k.cpu().numpy()
            lut = agg.strings.get(c)
            host.append(np.array([lut[int(x)] for x in h], dtype=object) if lut is not None else h)
        order = np.lexsort(tuple(reversed(host)))
        return torch.from_numpy(order).to(gkeys[0].device)
    order = torch.arange(gkeys[0].numel(), device=gkeys[0].device)
    for k in reversed(gkeys):
        order = order[torch.argsort(k[order], stable=True)]
    return order


def _finish_agg(agg: str, src_dtype, cnt, s, s2, mn, mx) -> DeviceColumn:
    """Per-group finishing arithmetic on the O(#groups) outputs of nvt_seg_aggregate -- pandas
    semantics: nulls are skipped; count -> int32; sum / mean / std / var -> float32."""
    if agg == "count":
        return DeviceColumn(cnt.to(torch.int32))
    if agg == "sum":
        return DeviceColumn(s.to(torch.float32))
    if agg == "mean":
        return DeviceColumn((s / cnt).to(torch.float32))
    if agg in ("std", "var"):
        var = (s2 - s * s / cnt) / (cnt - 1)
        var = torch.where(cnt > 1, var.clamp_min(0), torch.full_like(var, float("nan")))
        return DeviceColumn((var.sqrt() if agg == "std" else var).to(torch.float32))
    if agg in ("min", "max"):
        red = mn if agg == "min" else mx
        none = cnt == 0
        if src_dtype.is_floating_point:
            red = torch.where(none, torch.full_like(red, float("nan")), red)
            return DeviceColumn(red.to(src_dtype))
        vb = None if not bool(none.any()) else pack_bitmap_device(~none)
        return DeviceColumn(torch.where(none, torch.zeros_like(red), red).to(src_dtype), vb)
    raise NotImplementedError(agg)
