"""JoinGroupby (reference: nvtabular/ops/join_groupby.py:37-283)."""
from __future__ import annotations

import os

import numpy as np
import pandas as pd
import torch

from .. import kernels as K
from ..device import DeviceColumn, DeviceFrame, as_device_frame, key_view
from ..node import Node
from ..selector import ColumnSelector
from .base import StatOperator
from ._groupby import GroupAgg, derive_stats, stats_frame
from .categorify import _make_name

AGG_DTYPES = {"count": np.int32, "std": np.float32, "var": np.float32, "mean": np.float32}


class _LazyColumns:
    """{column name: float64 tensor [groups]} whose values are computed on first access (the names
    -- what schemas and the lookup plan need -- are known up front)."""

    def __init__(self, names, compute):
        self._names, self._compute, self._vals = list(names), compute, None

    def _get(self):
        if self._vals is None:
            self._vals = self._compute()
        return self._vals

    def __iter__(self):
        return iter(self._names)

    def __len__(self):
        return len(self._names)

    def __contains__(self, k):
        return k in self._names

    def __getitem__(self, k):
        return self._get()[k]

    def keys(self):
        return list(self._names)

    def values(self):
        return [self._get()[k] for k in self._names]

    def items(self):
        return [(k, self._get()[k]) for k in self._names]


class _Stats:
    """Device-resident stat table of one group: lookup index + stat columns."""

    def __init__(self, key_cols, keys, null_mask, columns, f32_columns=(), index_table=None):
        self.key_cols = key_cols
        # sum / min / max of a float32 column stay float32 (pandas' groupby result dtype)
        self.f32_columns = set(f32_columns)
        self.n = int(keys[0].numel()) if keys else 0
        if index_table is not None:
            # the fit table itself: compaction left every group's row number in its slot
            self.index = index_table
        else:
            self.index = K.GroupbyTable(len(key_cols), 0, max(64, 2 * self.n + 1))
            self.index.index_build([k.contiguous() for k in keys], null_mask)
        self.columns = columns  # name -> float64/int64 tensor [groups]
        self._records = None
        self._te_records = {}

    def records(self) -> torch.Tensor:
        """[groups, len(columns)] float64, one row per group in column order: the layout
        K.FlatIndex.gather reads with one probe per row."""
        if self._records is None:
            self._records = torch.stack([c.to(torch.float64) for c in self.columns.values()],
                                        dim=1).contiguous()
        return self._records

    def te_records(self, target) -> torch.Tensor:
        """[groups, 2] {sum, count} of one target (TargetEncoding without folds)."""
        if target not in self._te_records:
            self._te_records[target] = torch.stack(
                [self.columns[f"sum:{target}"].to(torch.float64),
                 self.columns["count"].to(torch.float64)], dim=1).contiguous()
        return self._te_records[target]


class JoinGroupby(StatOperator):
    def __init__(self, cont_cols=None, stats=("count",), split_out=None, split_every=None,
                 cat_cache="host", out_path=None, on_host=True, name_sep="_", tree_width=None,
                 defer_artifacts=False):
        super().__init__()
        # engine extension (as on Categorify): write the cat_stats.*.parquet files at
        # flush_artifacts() / Workflow.save() instead of inside fit
        self.defer_artifacts = defer_artifacts
        self._pending = {}
        self._hints = {}
        self.storage_name = {}
        self.name_sep = name_sep
        self.stats = stats
        self.split_out = split_out
        self.split_every = split_every
        self.out_path = out_path or "./"
        self.on_host = on_host
        self.cat_cache = cat_cache
        self.categories = {}
        self._device_stats = {}
        self._pending = {}
        self._cont_names = None
        if isinstance(cont_cols, Node):
            self.cont_cols = cont_cols
        elif isinstance(cont_cols, ColumnSelector):
            self.cont_cols = self._cont_names = cont_cols
        else:
            self.cont_cols = self._cont_names = ColumnSelector(cont_cols)
        supported = ["count", "sum", "mean", "std", "var", "min", "max"]
        for op in self.stats:
            if op not in supported:
                raise ValueError(op + " operation is not supported.")

    @property
    def cont_names(self):
        if self._cont_names:
            return self._cont_names
        if isinstance(self.cont_cols, Node) and self.cont_cols.output_schema:
            return self.cont_cols.output_columns
        if self._cont_names is not None:
            return self._cont_names
        raise RuntimeError(
            "Can't compute continuous columns used by `JoinGroupby` until `Workflow` is fit to "
            "dataset or schema."
        )

    def _group_list(self, col_selector):
        out = []
        for g in col_selector.grouped_names:
            cols = list(g) if isinstance(g, (tuple, list)) else [g]
            out.append((_make_name(*cols, sep=self.name_sep), cols))
        return out

    def fit_begin(self, col_selector):
        for group in col_selector.subgroups:
            if len(group.names) > 1:
                name = _make_name(*group.names, sep=self.name_sep)
                for col in group.names:
                    self.storage_name[col] = name
        sumsq = "std" in self.stats or "var" in self.stats
        minmax = "min" in self.stats or "max" in self.stats
        conts = list(self.cont_names.names)
        return {name: GroupAgg(name, cols, conts, sumsq=sumsq, minmax=minmax,
                               hint=self._hints.get(name, 0))
                for name, cols in self._group_list(col_selector)}

    def fit_partition(self, state, col_selector, df):
        frame, _ = as_device_frame(df)
        for agg in state.values():
            agg.update(frame)

    def fit_end(self, state, col_selector):
        base = os.path.join(self.out_path, "categories")
        os.makedirs(base, exist_ok=True)
        out = {}
        for name, agg in state.items():
            comp = agg.finalize()
            self._hints[name] = max(64, int(comp["n"]))
            d = os.path.join(base, f"cat_stats.{name}.parquet")
            self._pending[name] = (agg, comp, list(self.stats), d)
            if not self.defer_artifacts:
                self.flush_artifacts()
            out[name] = d
            # device cache for transform: the statistics as float64 columns, evaluated lazily --
            # the lookup image (K.jg_image) takes them straight from the accumulators; only the
            # column-wise paths (hashed index, NVT_LOOKUP_IMAGES=0) read these arrays
            spec = []   # (column name, statistic, value column index) in _bottom_level_groupby's order
            if "count" in self.stats:
                spec.append((f"{name}{self.name_sep}count", "count", 0))
            for j, cont in enumerate(agg.val_cols):
                # column order of _bottom_level_groupby's `required` list (categorify.py:1087-1131)
                for stat in ("sum", "mean", "min", "max", "var", "std"):
                    if stat in self.stats:
                        spec.append((f"{name}{self.name_sep}{cont}{self.name_sep}{stat}", stat, j))

            def compute(comp=comp, spec=spec):
                derived = derive_stats(comp, self.stats)
                return {cn: (comp["count"].to(torch.float64) if stat == "count" else derived[(j, stat)])
                        for cn, stat, j in spec}

            cols = _LazyColumns([c[0] for c in spec], compute)
            cols.spec, cols.comp = spec, comp
            f32 = [f"{name}{self.name_sep}{cont}{self.name_sep}{stat}" for cont in agg.val_cols
                   for stat in ("sum", "min", "max") if agg.val_dtypes.get(cont) == torch.float32]
            st = self._device_stats[name] = _Stats(agg.key_cols, comp["keys"], comp["null_mask"], cols,
                                                   f32_columns=f32, index_table=comp.get("index_table"))
            self._attach_image(name, st)
        return out

    def _plan(self, st):
        """(column, output dtype, value of a row without group) per statistic (join_groupby.py:
        29-34 AGG_DTYPES, :214 astype)."""
        plan = []
        for cname in st.columns:
            out_dt, miss = torch.float64, float("nan")
            if cname in st.f32_columns:
                out_dt = torch.float32
            for agg, npdt in AGG_DTYPES.items():
                if cname.endswith(f"{self.name_sep}{agg}"):
                    out_dt = torch.int32 if npdt == np.int32 else torch.float32
            plan.append((cname, out_dt, 0.0 if out_dt == torch.int32 else miss))
        return plan

    def _attach_image(self, name, st):
        """Sort-path groups: this operator's statistics, cast to their output dtypes, as a byte
        range of the key column's lookup image (K.FlatIndex.image_lookup)."""
        self._consumers = getattr(self, "_consumers", {})
        old = self._consumers.pop(name, None)
        if old is not None:
            old.release()
        plan = self._plan(st)
        if not (K.LOOKUP_IMAGES and isinstance(st.index, K.FlatIndex) and len(st.key_cols) == 1
                and 1 <= len(plan) <= 16):
            return
        # 8-byte values first: every value stays aligned to its size
        order = sorted(range(len(plan)), key=lambda i: 0 if plan[i][1] in (torch.float64, torch.int64) else 1)
        rel, at = {}, 0
        for i in order:
            rel[i] = at
            at += 8 if plan[i][1] in (torch.float64, torch.int64) else 4
        outputs = [(plan[i][0], plan[i][1], rel[i], False, plan[i][2]) for i in range(len(plan))]

        def fill(image, stride, offset, groups, st=st, plan=plan, rel=rel):
            spec = getattr(st.columns, "spec", None)
            if spec is not None:   # straight from the accumulators: one launch, no float64 columns
                K.jg_image(image, stride, st.columns.comp,
                           [(spec[i][1], spec[i][2], plan[i][1], offset + rel[i]) for i in range(len(plan))],
                           groups)
                return
            K.image_pack(image, stride, [(st.columns[plan[i][0]], plan[i][1], offset + rel[i])
                                         for i in range(len(plan))], groups)

        def parts(offset, groups, st=st, plan=plan, rel=rel):
            spec = getattr(st.columns, "spec", None)
            if spec is None:
                return []   # (statistics read from files: only fill() can write them)
            return [K.jg_image_part(st.columns.comp,
                                    [(spec[i][1], spec[i][2], plan[i][1], offset + rel[i])
                                     for i in range(len(plan))], groups)]

        has_spec = getattr(st.columns, "spec", None) is not None
        cons = K.LookupConsumer(self, name, at, outputs, fill, groups=st.n, parts=parts if has_spec else None)
        st.index.attach(cons)
        self._consumers[name] = cons

    def flush_artifacts(self):
        """Write any deferred cat_stats.<group>.parquet directories."""
        for agg, comp, stats, d in self._pending.values():
            os.makedirs(d, exist_ok=True)
            stats_frame(agg, comp, stats, self.name_sep).to_parquet(
                os.path.join(d, "part.0.parquet"), index=False)
        self._pending = {}

    def fit_finalize(self, dask_stats):
        for col in dask_stats:
            self.categories[col] = dask_stats[col]

    def _stats_for(self, name, cols):
        st = self._device_stats.get(name)
        if st is not None:
            return st
        df = pd.read_parquet(self.categories[name])
        st = _stats_from_frame(df, cols)
        self._device_stats[name] = st
        return st

    def transform(self, col_selector, df):
        frame, was_pandas = as_device_frame(df)
        new = DeviceFrame()
        for name, cols in self._group_list(col_selector):
            if not all(c in frame for c in cols):
                continue
            st = self._stats_for(name, cols)  # stat tables are stored under the group name
            keys, valids = [], []
            for c in cols:
                k, v = key_view(frame[c].materialize())
                keys.append(k)
                valids.append(v)
            plan = self._plan(st)  # (column, output dtype, value of a row without group)
            if isinstance(st.index, K.FlatIndex) and 1 <= len(plan) <= 16:
                cons = getattr(self, "_consumers", {}).get(name) if K.LOOKUP_IMAGES else None
                if cons is not None and cons in (getattr(st.index, "consumers", None) or []):
                    # ONE probe + ONE packed record per row for every operator fitted on this
                    # key column in the same pass (TargetEncoding's values ride in the same launch)
                    got, unseen = st.index.image_lookup(cons, keys, valids)
                    outs = [got[p[0]] for p in plan]
                else:
                    # sort-path groups: probe + every statistic of the group's record in ONE launch
                    outs, unseen = st.index.gather(keys, valids, st.records(), [p[1] for p in plan],
                                                   [p[2] for p in plan])
                any_unseen = None  # read back once, and only when an integer column needs it
                for (cname, out_dt, _), o in zip(plan, outs):
                    if cname in new:
                        continue
                    if out_dt == torch.int32 and any_unseen is None:
                        any_unseen = bool(int(K.read_back(unseen)[0]))
                    if out_dt == torch.int32 and any_unseen:
                        raise ValueError(
                            f"Cannot convert non-finite values (NA or inf) to integer: column "
                            f"{cname} has unseen categories"
                        )
                    new[cname] = DeviceColumn(o)
                continue
            grp = st.index.lookup(keys, valids)
            for cname, src in st.columns.items():
                if cname in new:
                    continue
                out_dt, miss = torch.float64, float("nan")
                if cname in st.f32_columns:
                    out_dt = torch.float32
                for agg, npdt in AGG_DTYPES.items():
                    if cname.endswith(f"{self.name_sep}{agg}"):
                        out_dt = torch.int32 if npdt == np.int32 else torch.float32
                if out_dt == torch.int32:
                    if bool((grp < 0).any()):
                        # join_groupby.py:214 astype(int32) on NaN raises in the reference too
                        raise ValueError(
                            f"Cannot convert non-finite values (NA or inf) to integer: column "
                            f"{cname} has unseen categories"
                        )
                    miss = 0.0
                new[cname] = DeviceColumn(K.gather(src, grp, miss, out_dt))
        return new.to_pandas() if was_pandas else new

    @property
    def dependencies(self):
        return self.cont_cols  # selector or node: either way an upstream dependency

    def compute_selector(self, input_schema, selector, parents_selector=None,
                         dependencies_selector=None):
        self._validate_matching_cols(input_schema, parents_selector, "computing input selector")
        return parents_selector

    def column_mapping(self, col_selector):
        mapping = {}
        for group in col_selector.grouped_names:
            if isinstance(group, (tuple, list)):
                name = _make_name(*group, sep=self.name_sep)
                group = [*group]
            else:
                name = group
                group = [group]
            for cont in self.cont_names.names:
                for stat in self.stats:
                    if stat == "count":
                        mapping[f"{name}_{stat}"] = [*group]
                    else:
                        mapping[f"{name}_{cont}_{stat}"] = [cont, *group]
        return mapping

    def _compute_dtype(self, col_schema, input_schema):
        new_schema = super()._compute_dtype(col_schema, input_schema)
        dtype = new_schema.dtype
        for agg in AGG_DTYPES:
            if new_schema.name.endswith(f"{self.name_sep}{agg}"):
                dtype = AGG_DTYPES[agg]
                break
        return new_schema.with_dtype(dtype, is_list=False, is_ragged=False)

    def set_storage_path(self, new_path, copy=False):
        import shutil

        self.flush_artifacts()
        new = {}
        for col, old in self.categories.items():
            target = old.replace(str(self.out_path), str(new_path))
            if copy and target != old:
                shutil.copytree(old, target, dirs_exist_ok=True)
            new[col] = target
        self.categories = new
        self.out_path = new_path

    def prepare_transform(self):
        """Called by Workflow.fit once every operator is fitted: the lookup images this operator
        shares with the other operators on its key columns are enqueued now."""
        for cons in getattr(self, "_consumers", {}).values():
            if cons.index is not None:
                cons.index.prepare_image()

    def clear(self):
        self.categories = {}
        self.storage_name = {}
        self._device_stats = {}
        for cons in getattr(self, "_consumers", {}).values():
            cons.release()
        self._consumers = {}
        self._pending = {}


def _stats_from_frame(df: pd.DataFrame, key_cols) -> _Stats:
    """Rebuild the device stat table from a cat_stats parquet frame."""
    from ..strings import string_key64

    dev = torch.device("cuda", torch.cuda.current_device())
    keys, nm = [], np.zeros(len(df), dtype=np.uint8)
    for j, c in enumerate(key_cols):
        s = df[c]
        isnull = s.isna().to_numpy()
        if s.dtype == object or pd.api.types.is_string_dtype(s.dtype):
            hk = np.zeros(len(s), dtype=np.int64)
            if (~isnull).any():
                hk[~isnull] = string_key64(s.to_numpy(dtype=object)[~isnull])
        else:
            hk = s.fillna(0).to_numpy().astype(np.int64)
        nm |= isnull.astype(np.uint8) << j
        keys.append(torch.from_numpy(hk).to(dev))
    cols = {
        c: torch.from_numpy(df[c].to_numpy().astype(np.float64)).to(dev)
        for c in df.columns if c not in key_cols
    }
    f32 = [c for c in cols if df[c].dtype == np.float32 and c.rsplit("_", 1)[-1] in ("sum", "min", "max")]
    return _Stats(list(key_cols), keys, torch.from_numpy(nm).to(dev), cols, f32_columns=f32)
