"""TargetEncoding (reference: nvtabular/ops/target_encoding.py:30-439)."""
from __future__ import annotations

import os

import numpy as np
import pandas as pd
import torch

from .. import kernels as K
from ..device import DeviceColumn, DeviceFrame, as_device_frame, key_view
from ..node import Node
from ..schema import Tags
from ..selector import ColumnSelector
from .base import StatOperator
from ._groupby import GroupAgg, fold_sparse, stats_frame
from .categorify import _make_name
from .join_groupby import _Stats, _stats_from_frame
from .normalize import PendingMoments, moments_begin, moments_end_async, moments_partition


def _add_fold(n, kfold, fold_seed=None) -> np.ndarray:
    """target_encoding.py:427-439: per-partition fold ids (host RNG so that the
    sequence is numpy's MT19937 stream, bit-identical to the reference; 1 B/row)."""
    typ = np.min_scalar_type(kfold * 2)
    if fold_seed is None:
        fold = np.arange(n, dtype=typ)
        np.mod(fold, kfold, out=fold)
        return fold
    state = np.random.RandomState(fold_seed)
    return state.choice(np.arange(kfold, dtype=typ), n)


_FOLD_CACHE = {}  # (kfold, fold_seed, device) -> uint8 tensor of the longest partition seen
DEVICE_FOLDS = os.environ.get("NVT_DEVICE_FOLDS", "1") != "0"
PARALLEL_FOLDS = os.environ.get("NVT_PARALLEL_FOLDS", "1") != "0"
PARALLEL_FOLDS_MIN = 1 << 19   # below: one workgroup walks the generator (a few ms)


def _fold_column(n, kfold, fold_seed, device) -> DeviceColumn:
    """Fold ids of one partition in HBM.

    fold_seed None: ``arange(n) % kfold`` is generated on the device (no host traffic).
    Seeded folds must be numpy's MT19937 stream to stay bit-identical to the reference
    (target_encoding.py:436-439): regenerated on the device (nvt_fold_mt19937: init_genrand +
    masked rejection, bit-checked against numpy in tests/test_gpu_sorted_groupby.py; the host
    draws them only for seeds / kfold outside the kernel's range).  The reference re-seeds per
    partition, i.e. every partition gets the SAME sequence (a shorter partition a prefix of it):
    the column is generated once per (kfold, seed, device) and sliced afterwards."""
    if fold_seed is None:
        # the reference's arange is created IN the fold dtype (min_scalar_type(kfold * 2), uint8
        # for any sane kfold) and wraps: fold = (i mod 2^bits) mod kfold, not i mod kfold
        bits = 8 * np.dtype(np.min_scalar_type(kfold * 2)).itemsize
        idx = torch.arange(n, device=device, dtype=torch.int64)
        if bits < 64:
            idx = idx & ((1 << bits) - 1)
        return DeviceColumn((idx % kfold).to(torch.uint8 if kfold <= 255 else torch.int32))
    key = (int(kfold), int(fold_seed), str(device))
    cached = _FOLD_CACHE.get(key)
    if cached is None or cached.numel() < n:
        seed = int(fold_seed)
        if DEVICE_FOLDS and 1 <= kfold <= 128 and 0 <= seed < 2**32 and torch.device(device).type == "cuda":
            # numpy's stream regenerated on the device (nvt_fold_mt19937): no O(rows) host work
            # and no copy; (a little head room: a longer partition later extends, not restarts)
            m = int(n) + int(n) // 16
            cached = torch.empty(m, dtype=torch.uint8, device=device)
            lib = K._lib.load()
            done = False
            if PARALLEL_FOLDS and m >= PARALLEL_FOLDS_MIN:
                # chunks of 2^18 draws generated in parallel (jump-ahead), then put behind each
                # other; the chunk count carries an 8-sigma margin -- the total is checked
                import ctypes as C

                need = C.c_uint64()
                try:
                    # (a request the parallel generator rejects -- more than 2^14 chunks -- or a
                    # workspace that cannot be had falls through to the serial kernel)
                    K.check(lib.nvt_fold_mt19937_par_ws_bytes(m, int(kfold), C.byref(need)),
                            "nvt_fold_mt19937_par_ws_bytes")
                    ws = torch.empty(need.value, dtype=torch.uint8, device=device)
                    total = torch.zeros(1, dtype=torch.int64, device=device)
                    K.check(lib.nvt_fold_mt19937_par(seed, int(kfold), m, cached.data_ptr(), ws.data_ptr(),
                                                     need.value, total.data_ptr(), K.stream_ptr()),
                            "nvt_fold_mt19937_par")
                    done = int(K.read_back(total)[0]) >= m
                    del ws
                except (K._lib.NvtHipError, torch.cuda.OutOfMemoryError):
                    done = False
            if not done:
                K.check(lib.nvt_fold_mt19937(seed, int(kfold), m, cached.data_ptr(), K.stream_ptr()),
                        "nvt_fold_mt19937")
        else:
            f = _add_fold(n, kfold, fold_seed).astype(np.uint8)
            cached = torch.from_numpy(f).to(device)
        _FOLD_CACHE[key] = cached
    return DeviceColumn(cached[:n])


class _FoldDense:
    """Per-(group, fold) statistics of the sort path: entry g * kfold + fold of ``count`` /
    ``sums[target]`` (K.sorted_groupby); the transform needs no second lookup."""

    def __init__(self, kfold, count, sums, records=None):
        self.kfold, self.count, self.sums = int(kfold), count.contiguous(), sums
        # target -> [groups, 2 * (kfold + 1)] {sum, count, (sum_f, count_f) ...}: the one-probe
        # layout of K.FlatIndex.te
        self.records = records


class TargetEncoding(StatOperator):
    def __init__(self, target, target_mean=None, kfold=None, fold_seed=42, p_smooth=20,
                 out_col=None, out_dtype=None, split_out=None, split_every=None,
                 cat_cache="host", out_path=None, on_host=True, name_sep="_", drop_folds=True,
                 tree_width=None, defer_artifacts=False):
        super().__init__()
        self.defer_artifacts = defer_artifacts  # engine extension, see JoinGroupby
        self._pending = {}
        self._hints = {}
        target = Node.construct_from(target)
        self.dependency = target
        self.target = target
        self.target_mean = target_mean
        self.kfold = kfold or 3
        self.fold_seed = fold_seed
        self.p_smooth = p_smooth
        self.out_col = [out_col] if isinstance(out_col, str) else out_col
        self.out_dtype = out_dtype
        self.split_out = split_out
        self.split_every = split_every
        self.out_path = out_path or "./"
        self.on_host = on_host
        self.cat_cache = cat_cache
        self.name_sep = name_sep
        self.drop_folds = drop_folds
        self.fold_name = "__fold__"
        self.stats = {}
        self.means = {}
        self._device_stats = {}

    # ------------------------------------------------------------------ fit --
    def _groups(self, col_selector):
        out = []
        for g in col_selector.grouped_names:
            out.append(list(g) if isinstance(g, (tuple, list)) else [g])
        return out

    def fit_begin(self, col_selector):
        targets = list(self.target_columns)
        state = {"moments": None, "aggs": {}}
        if self.target_mean is None:
            state["moments"] = moments_begin(targets)
        for cols in self._groups(col_selector):
            name = _make_name(*cols, sep=self.name_sep)
            if self.kfold > 1:
                # the [fold, key] aggregate rides on the key aggregate (one sort serves both
                # when the key is one int32 column; two hash tables otherwise)
                fcols = [self.fold_name] + cols
                fname = _make_name(*fcols, sep=self.name_sep)
                state["aggs"][name] = GroupAgg(name, cols, targets, hint=self._hints.get(name, 0),
                                               fold=(self.fold_name, self.kfold), fold_name=fname,
                                               fold_hint=self._hints.get(fname, 0))
            else:
                state["aggs"][name] = GroupAgg(name, cols, targets, hint=self._hints.get(name, 0))
        return state

    def fit_partition(self, state, col_selector, df):
        frame, _ = as_device_frame(df)
        if state["moments"] is not None:
            moments_partition(state["moments"], frame)
        if self.kfold > 1 and self.fold_name not in frame:
            frame = frame.copy()
            dev = next(iter(frame.items()))[1].data.device
            frame[self.fold_name] = _fold_column(len(frame), self.kfold, self.fold_seed, dev)
        for agg in state["aggs"].values():
            agg.update(frame)

    def fit_end(self, state, col_selector):
        base = os.path.join(self.out_path, "categories")
        os.makedirs(base, exist_ok=True)
        paths = {}

        def register(name, agg, comp):
            self._hints[name] = max(64, int(comp["n"]))
            d = os.path.join(base, f"cat_stats.{name}.parquet")
            self._pending[name] = (agg, comp, d)
            paths[name] = d
            cols = {"count": comp["count"]}
            for j, t in enumerate(agg.val_cols):
                cols[f"sum:{t}"] = comp["sum"][j]
            self._device_stats[name] = _Stats(agg.key_cols, comp["keys"], comp["null_mask"], cols,
                                              index_table=comp.get("index_table"))

        for name, agg in state["aggs"].items():
            comp = agg.finalize()
            register(name, agg, comp)
            if agg.fold is None:
                continue
            fname = agg.fold_name
            if "fold" not in comp:  # two hash tables
                fa = agg._fold_classic()
                register(fname, fa, fa.finalize())
                continue
            # sort path: dense per-(group, fold) statistics; the [fold, key] groups of the
            # artifact are derived from them when the file is written
            f = comp["fold"]
            view = GroupAgg(fname, [self.fold_name] + agg.key_cols, agg.val_cols)
            view.key_dtypes = {self.fold_name: torch.uint8, **agg.key_dtypes}
            view.val_dtypes = dict(agg.val_dtypes)
            self._hints[fname] = max(64, int(comp["n"]) * min(self.kfold, 2))
            d = os.path.join(base, f"cat_stats.{fname}.parquet")
            self._pending[fname] = (view, (lambda c=comp: fold_sparse(c)), d)
            paths[fname] = d
            self._device_stats[fname] = _FoldDense(
                self.kfold, f["size"], {t: f["sum"][j] for j, t in enumerate(agg.val_cols)},
                records=({t: f["records"][j] for j, t in enumerate(agg.val_cols)}
                         if f.get("records") is not None else None))
        if not self.defer_artifacts:
            self.flush_artifacts()
        # the target means start their way to the host here and are waited for when something
        # needs them as host numbers (prepare_transform at the end of Workflow.fit, or the first
        # reader of self.means): the lookup image takes sum / count from the device, so that it is
        # enqueued while the device still works on this fit
        moments = moments_end_async(state["moments"]) if state["moments"] is not None else None
        return paths, moments

    def flush_artifacts(self):
        """Write any deferred cat_stats.<group>.parquet directories."""
        for agg, comp, d in self._pending.values():
            if callable(comp):
                comp = comp()
            os.makedirs(d, exist_ok=True)
            stats_frame(agg, comp, ["count", "sum"], self.name_sep).to_parquet(
                os.path.join(d, "part.0.parquet"), index=False)
        self._pending = {}

    def fit_finalize(self, dask_stats):
        for col, value in dask_stats[0].items():
            self.stats[col] = value
        self._pending_moments = None
        if isinstance(dask_stats[1], PendingMoments):
            self._pending_moments = dask_stats[1]
        elif dask_stats[1] is not None:
            for col, m in dask_stats[1].items():
                self._means[col] = float(m["mean"])
        self._attach_images()

    @property
    def means(self):
        """{target: mean of the fit} (resolves the read-back a fit left pending)."""
        pm = getattr(self, "_pending_moments", None)
        if pm is not None:
            self._pending_moments = None
            for col, m in pm.resolve().items():
                self._means[col] = float(m["mean"])
        return self._means

    @means.setter
    def means(self, value):
        self._pending_moments = None
        self._means = value

    def _out_torch_dtype(self):
        return torch.float64 if np.dtype(self.output_dtype) == np.dtype("float64") else torch.float32

    def _attach_images(self):
        """Sort-path groups (one int32 / narrow int64 key column): this operator's byte range of
        the key column's lookup image (K.FlatIndex.image_lookup) -- (kfold + 1) smoothed values
        per target and group, target_encoding.py:350-371 evaluated once per (group, fold)."""
        for old in getattr(self, "_consumers", {}).values():
            old.release()
        self._consumers = {}
        if not K.LOOKUP_IMAGES:
            return
        fit_folds = self.kfold > 1
        # means still on their way to the host (fit_end): the one-pass image build reads sum /
        # count from the device; whatever needs the number itself waits for it then
        pm = getattr(self, "_pending_moments", None) if self.target_mean is None else None
        dev_moments = {t: pm.acc[i] for i, t in enumerate(pm.names)} if pm is not None else {}
        out_dt = self._out_torch_dtype()
        size = 8 if out_dt == torch.float64 else 4
        slots = (self.kfold + 1) if fit_folds else 1

        def mean_of(t):
            y_mean = self.target_mean or self.means
            return float(y_mean[t] if isinstance(y_mean, dict) else y_mean)

        for name, st_all in list(self._device_stats.items()):
            if not isinstance(st_all, _Stats) or not isinstance(st_all.index, K.FlatIndex):
                continue
            if len(st_all.key_cols) != 1:
                continue
            st_fold = None
            if fit_folds:
                fname = _make_name(self.fold_name, *st_all.key_cols, sep=self.name_sep)
                st_fold = self._device_stats.get(fname)
                if not isinstance(st_fold, _FoldDense):
                    continue
            targets = [c[len("sum:"):] for c in st_all.columns if c.startswith("sum:")]
            if pm is None:
                try:
                    for t in targets:
                        mean_of(t)
                except (KeyError, TypeError):
                    continue
            elif any(t not in dev_moments for t in targets):
                continue
            # (the value of a row without group: a callable, asked for when a lookup is launched)
            outputs = [(("te", t), out_dt, j * slots * size, fit_folds, (lambda t=t: mean_of(t)))
                       for j, t in enumerate(targets)]

            def fill(image, stride, offset, groups, st_all=st_all, st_fold=st_fold, targets=targets):
                cnt = st_all.columns["count"].to(torch.int64)
                for j, t in enumerate(targets):
                    K.te_image(image, stride, offset + j * slots * size, cnt,
                               st_all.columns[f"sum:{t}"].to(torch.float64),
                               st_fold.count if fit_folds else None,
                               st_fold.sums[t] if fit_folds else None,
                               self.kfold if fit_folds else 0, groups, self.p_smooth, mean_of(t), out_dt)

            def parts(offset, groups, st_all=st_all, st_fold=st_fold, targets=targets):
                cnt = st_all.columns["count"].to(torch.int64)
                return [K.te_image_part(offset + j * slots * size, cnt,
                                        st_all.columns[f"sum:{t}"].to(torch.float64),
                                        st_fold.count if fit_folds else None,
                                        st_fold.sums[t] if fit_folds else None,
                                        self.kfold if fit_folds else 0, groups, self.p_smooth,
                                        0.0 if t in dev_moments else mean_of(t), out_dt,
                                        moments=dev_moments.get(t))
                        for j, t in enumerate(targets)]

            fold_fn = None
            if fit_folds:
                fold_fn = lambda n, dev: _fold_column(n, self.kfold, self.fold_seed, dev).data  # noqa: E731
            cons = K.LookupConsumer(self, name, len(targets) * slots * size, outputs, fill, fold_fn,
                                    groups=st_all.n, parts=parts)
            st_all.index.attach(cons)
            self._consumers[name] = cons

    # ------------------------------------------------------------ transform --
    def _stats_for(self, name, key_cols):
        st = self._device_stats.get(name)
        if st is not None:
            return st
        df = pd.read_parquet(self.stats[name])
        targets = list(self.target_columns)
        # file layout (categorify.py:1079-1137): keys, <name>_count, <name>_<t>_sum...
        ren = {f"{name}{self.name_sep}count": "count"}
        for t in targets:
            ren[f"{name}{self.name_sep}{t}{self.name_sep}sum"] = f"sum:{t}"
        st = _stats_from_frame(df.rename(columns=ren), key_cols)
        st.columns["count"] = st.columns["count"].to(torch.int64)
        self._device_stats[name] = st
        return st

    def transform(self, col_selector, df):
        frame, was_pandas = as_device_frame(df)
        fit_folds = self.kfold > 1
        dev = next(iter(frame.items()))[1].data.device
        work = frame
        if fit_folds:
            work = frame.copy()
            work[self.fold_name] = _fold_column(len(frame), self.kfold, self.fold_seed, dev)
        y_mean = self.target_mean or self.means
        targets = list(self.target_columns)
        out_dt = torch.float64 if np.dtype(self.output_dtype) == np.dtype("float64") else torch.float32
        new = DeviceFrame()
        for ind, cat_group in enumerate(self._groups(col_selector)):
            if isinstance(self.out_col, list):
                if ind >= len(self.out_col):
                    raise ValueError("out_col and cat_groups are different sizes.")
                out_col = self.out_col[ind]
                out_col = [out_col] if isinstance(out_col, str) else out_col
                if len(out_col) != len(targets):
                    raise ValueError("out_col and target are different sizes.")
            else:
                tag = _make_name(*cat_group, sep=self.name_sep)
                out_col = [f"TE_{tag}_{x}" for x in targets]
            name_all = _make_name(*cat_group, sep=self.name_sep)
            st_all = self._stats_for(name_all, cat_group)

            def lookup(st, cols):
                keys, valids = [], []
                for c in cols:
                    k, v = key_view(work[c].materialize())
                    keys.append(k)
                    valids.append(v)
                return st.index.lookup(keys, valids)

            g_fold, st_fold = None, None
            if fit_folds:
                fcols = [self.fold_name] + cat_group
                st_fold = self._stats_for(_make_name(*fcols, sep=self.name_sep), fcols)
            # one int32 key column fitted on the sort path: probe + formula in ONE launch per
            # target (no group-id columns in HBM)
            fused = isinstance(st_all.index, K.FlatIndex) and (
                (not fit_folds) or (isinstance(st_fold, _FoldDense) and st_fold.records is not None))
            cons = None
            if K.LOOKUP_IMAGES and isinstance(st_all.index, K.FlatIndex):
                cons = getattr(self, "_consumers", {}).get(name_all)
            if cons is not None and getattr(st_all.index, "consumers", None) and cons in st_all.index.consumers \
                    and [o[0][1] for o in cons.outputs] == targets and cons.outputs[0][1] == out_dt:
                # ONE probe + ONE packed record per row for every operator fitted on this key
                # column in the same pass (JoinGroupby's statistics ride in the same launch)
                k, v = key_view(work[cat_group[0]].materialize())
                fold_t = work[self.fold_name].data if fit_folds else None
                got, _ = st_all.index.image_lookup(cons, [k], [v], fold=fold_t)
                for i, t in enumerate(targets):
                    new[out_col[i]] = DeviceColumn(got[("te", t)])
                continue
            if fused:
                k, v = key_view(work[cat_group[0]].materialize())
                for i, t in enumerate(targets):
                    ym = y_mean[t] if isinstance(y_mean, dict) else y_mean
                    if fit_folds:
                        rec, fold_t = st_fold.records[t], work[self.fold_name].data
                    else:
                        rec, fold_t = st_all.te_records(t), None
                    new[out_col[i]] = DeviceColumn(st_all.index.te(
                        [k], [v], fold_t, self.kfold, rec, self.p_smooth, ym, out_dt))
                continue
            g_all = lookup(st_all, cat_group)
            if fit_folds and not isinstance(st_fold, _FoldDense):
                g_fold = lookup(st_fold, fcols)
            for i, t in enumerate(targets):
                ym = y_mean[t] if isinstance(y_mean, dict) else y_mean
                if isinstance(st_fold, _FoldDense):
                    new[out_col[i]] = DeviceColumn(K.te_apply_folds(
                        g_all, work[self.fold_name].data, st_fold.kfold,
                        st_all.columns[f"sum:{t}"].to(torch.float64).contiguous(),
                        st_all.columns["count"].to(torch.int64).contiguous(),
                        st_fold.sums[t], st_fold.count, self.p_smooth, ym, out_dt))
                    continue
                out = K.te_apply(
                    g_all, g_fold,
                    st_all.columns[f"sum:{t}"].to(torch.float64).contiguous(),
                    st_all.columns["count"].to(torch.int64).contiguous(),
                    st_fold.columns[f"sum:{t}"].to(torch.float64).contiguous() if fit_folds else None,
                    st_fold.columns["count"].to(torch.int64).contiguous() if fit_folds else None,
                    self.p_smooth, ym, out_dt,
                )
                new[out_col[i]] = DeviceColumn(out)
        if fit_folds and not self.drop_folds:
            new[self.fold_name] = work[self.fold_name]
        return new.to_pandas() if was_pandas else new

    # --------------------------------------------------------------- schema --
    @property
    def dependencies(self):
        return self.dependency

    def compute_selector(self, input_schema, selector, parents_selector=None,
                         dependencies_selector=None):
        self._validate_matching_cols(input_schema, parents_selector, "computing input selector")
        return parents_selector

    def column_mapping(self, col_selector):
        mapping = {}
        for group in col_selector.grouped_names:
            group = ColumnSelector(group if isinstance(group, str) else list(group))
            tag = _make_name(*group.names, sep=self.name_sep)
            for target_name in self.target_columns:
                mapping[f"TE_{tag}_{target_name}"] = [target_name, *group.names]
        if self.kfold > 1 and not self.drop_folds:
            mapping[self.fold_name] = []
        return mapping

    def _compute_dtype(self, col_schema, input_schema):
        if input_schema.column_schemas:
            return super()._compute_dtype(col_schema, input_schema).with_dtype(
                self.output_dtype, is_list=False, is_ragged=False)
        return col_schema.with_dtype(np.uint8)

    def _compute_tags(self, col_schema, input_schema):
        if input_schema.column_schemas:
            src = input_schema.column_names[0]
            return col_schema.with_tags(list(input_schema[src].tags) + self.output_tags)
        return col_schema

    @property
    def output_dtype(self):
        return self.out_dtype or np.float32

    @property
    def output_tags(self):
        return [Tags.CONTINUOUS]

    @property
    def target_columns(self):
        if self.target.output_schema is not None:
            return self.target.output_schema.column_names
        if self.target.selector is not None:
            return self.target.selector.names
        return []

    def set_storage_path(self, new_path, copy=False):
        import shutil

        self.flush_artifacts()
        new = {}
        for col, old in self.stats.items():
            target = old.replace(str(self.out_path), str(new_path))
            if copy and target != old:
                shutil.copytree(old, target, dirs_exist_ok=True)
            new[col] = target
        self.stats = new
        self.out_path = new_path

    def prepare_transform(self):
        """Called by Workflow.fit once every operator is fitted: the lookup images this operator
        shares with the other operators on its key columns are enqueued now."""
        for cons in getattr(self, "_consumers", {}).values():
            if cons.index is not None:
                cons.index.prepare_image()
        _ = self.means   # (the fit's means reach the host here, behind the image's launches)

    def clear(self):
        self.stats = {}
        self.means = {}
        self._device_stats = {}
        for cons in getattr(self, "_consumers", {}).values():
            cons.release()
        self._consumers = {}
        self._pending = {}
