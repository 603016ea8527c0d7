"""Normalize / NormalizeMinMax (reference: nvtabular/ops/normalize.py, moments.py).

fit  = one streaming pass per partition accumulating {count, sum, sum of squares}
       per column in fp64 (``nvt_moments``); partitions / GPUs combine by addition
       (moments.py:80-86 tree-sum; RCCL all-reduce across ranks).
transform = ``nvt_fill_normalize``: (x - mean) / std in fp64, fused with a pending
       FillMissing constant, written as float64 (or ``out_dtype``).
"""
from __future__ import annotations

import math

import numpy
import torch

from .. import kernels as K
from ..device import DeviceColumn, as_device_frame, torch_dtype
from ..schema import Tags
from ..selector import ColumnSelector
from .base import StatOperator


def finalize_moments(count, total, sq, ddof=1):
    """moments.py:89-116 for one column -> (mean, var, std)."""
    n = count
    if n == 0:
        return float("nan"), float("nan"), float("nan")
    var = sq - total * total / n
    div = n - ddof
    if div < 1:
        div = 1
    var = var / div
    if (n - ddof) == 0:
        var = float("nan")
    mean = total / n
    std = math.sqrt(var) if var == var and var >= 0 else float("nan")
    return mean, var, std


class _MomentState:
    def __init__(self, names):
        self.names = list(names)
        self.acc = None  # float64 [ncols, 3] on device


def moments_begin(names):
    return _MomentState(names)


def moments_partition(state: _MomentState, frame):
    cols = [frame[name] for name in state.names]
    if state.acc is None and cols:
        state.acc = torch.zeros(len(state.names), 3, dtype=torch.float64, device=cols[0].data.device)
    # every column of the partition in ONE launch (nvt_moments_many)
    K.moments_many([(c.data, c.valid, c.fill, state.acc[i]) for i, c in enumerate(cols)])


def moments_end(state: _MomentState):
    """{name: dict(count,sum,sum2,mean,var,std)} after the cross-rank reduction."""
    from .. import dist

    acc = state.acc
    if acc is None:
        acc = torch.zeros(len(state.names), 3, dtype=torch.float64,
                          device=torch.device("cuda", torch.cuda.current_device()))
    acc = dist.all_reduce_sum(acc)
    host = K.read_back(acc).tolist()
    out = {}
    for name, (n, s, s2) in zip(state.names, host):
        mean, var, std = finalize_moments(n, s, s2)
        out[name] = dict(count=n, sum=s, sum2=s2, mean=mean, var=var, std=std)
    return out


class PendingMoments:
    """moments_end in two halves: the accumulators are on their way to the host when this is
    returned, ``resolve()`` waits and finishes them.  ``acc`` ([names, 3] float64 {count, sum,
    sum of squares}, already reduced over the ranks) stays on the device for kernels that take
    the mean from there (sum / count: the same IEEE division finalize_moments makes)."""

    def __init__(self, names, acc):
        self.names, self.acc = list(names), acc
        self._pending = K.PendingReadBack(acc)
        self._out = None

    def resolve(self):
        if self._out is None:
            out = {}
            for name, (n, s, s2) in zip(self.names, self._pending.get().tolist()):
                mean, var, std = finalize_moments(n, s, s2)
                out[name] = dict(count=n, sum=s, sum2=s2, mean=mean, var=var, std=std)
            self._out = out
        return self._out


def moments_end_async(state: _MomentState) -> PendingMoments:
    from .. import dist

    acc = state.acc
    if acc is None:
        acc = torch.zeros(len(state.names), 3, dtype=torch.float64,
                          device=torch.device("cuda", torch.cuda.current_device()))
    return PendingMoments(state.names, dist.all_reduce_sum(acc))


class Normalize(StatOperator):
    """Standardise continuous columns with the mean/std method (normalize.py:33-124)."""

    def __init__(self, out_dtype=None):
        super().__init__()
        self._pending = None
        self.means = {}
        self.stds = {}
        self.out_dtype = out_dtype

    def fit_begin(self, col_selector: ColumnSelector):
        return moments_begin(col_selector.names)

    def fit_partition(self, state, col_selector, df):
        frame, _ = as_device_frame(df)
        moments_partition(state, frame)

    def fit_end(self, state, col_selector):
        # the accumulators start their way to the host; nobody waits here.  A transform that
        # follows at once hands the kernel the device accumulators (it finishes mean / std as
        # finalize_moments does) and is enqueued while the fit's last kernels still run; the host
        # numbers are there for whoever reads means / stds.
        return moments_end_async(state)

    def fit_finalize(self, stats):
        if isinstance(stats, PendingMoments):
            self._means, self._stds, self._pending = {}, {}, stats
            return
        self._pending = None
        for col, m in stats.items():
            self._means[col] = float(m["mean"])
            self._stds[col] = float(m["std"])

    def _resolve(self):
        pm, self._pending = self._pending, None
        if pm is not None:
            for col, m in pm.resolve().items():
                self._means[col] = float(m["mean"])
                self._stds[col] = float(m["std"])

    @property
    def means(self):
        self._resolve()
        return self._means

    @means.setter
    def means(self, value):
        self._pending = None
        self._means = value

    @property
    def stds(self):
        self._resolve()
        return self._stds

    @stds.setter
    def stds(self, value):
        self._pending = None
        self._stds = value

    def transform(self, col_selector: ColumnSelector, df):
        frame, was_pandas = as_device_frame(df)
        from ..device import DeviceFrame

        new = DeviceFrame()
        out_dt = torch_dtype(self.output_dtype)
        items, cols = [], []
        pm = self._pending
        dev = {n: pm.acc[i] for i, n in enumerate(pm.names)} if pm is not None else {}
        if any(n not in dev for n in col_selector.names):
            dev = {}
            self._resolve()
        for name in col_selector.names:
            col = frame[name]
            data = col.data.view(torch.uint8) if col.data.dtype == torch.bool else col.data
            if dev:   # (the fit's moments have not been asked for on the host yet)
                items.append((data, col.valid, col.fill, True, 0.0, 0.0, out_dt, False, dev[name]))
            else:
                std = self._stds[name]
                scale = std if std > 0 else 0.0  # normalize.py:79-82: std == 0 -> x - mean
                items.append((data, col.valid, col.fill, True, self._means[name], scale, out_dt, False))
            cols.append((name, col))
        # FillMissing + Normalize of every column in ONE launch (nvt_fill_normalize_many)
        for (name, col), (out, _) in zip(cols, K.fill_normalize_many(items)):
            new[name] = DeviceColumn(out, None, col.offsets)
        if dev:
            self._resolve()   # (behind the launch: the device has work while the host waits)
        return new.to_pandas() if was_pandas else new

    def clear(self):
        self.means = {}
        self.stds = {}

    @property
    def output_tags(self):
        return [Tags.CONTINUOUS]

    @property
    def output_dtype(self):
        return self.out_dtype or numpy.float64


class NormalizeMinMax(StatOperator):
    """(x - min) / (max - min) (normalize.py:127-212)."""

    def __init__(self, out_dtype=None):
        super().__init__()
        self.mins = {}
        self.maxs = {}
        self.out_dtype = out_dtype

    def fit_begin(self, col_selector):
        return {"names": list(col_selector.names), "acc": None, "first": True}

    def fit_partition(self, state, col_selector, df):
        frame, _ = as_device_frame(df)
        for i, name in enumerate(state["names"]):
            col = frame[name].materialize()
            if state["acc"] is None:
                state["acc"] = torch.full((len(state["names"]), 2), float("nan"),
                                          dtype=torch.float64, device=col.data.device)
            K.minmax_accumulate(col.data, col.valid, state["acc"][i], first=False)

    def fit_end(self, state, col_selector):
        from .. import dist

        acc = state["acc"]
        mn = dist.all_reduce_min(acc[:, 0].contiguous())
        mx = dist.all_reduce_max(acc[:, 1].contiguous())
        return {n: (float(a), float(b)) for n, a, b in zip(state["names"], mn.cpu().tolist(),
                                                             mx.cpu().tolist())}

    def fit_finalize(self, stats):
        for col, (mn, mx) in stats.items():
            self.mins[col] = mn
            self.maxs[col] = mx

    def transform(self, col_selector, df):
        frame, was_pandas = as_device_frame(df)
        from ..device import DeviceFrame

        new = DeviceFrame()
        out_dt = torch_dtype(self.output_dtype)
        for name in col_selector.names:
            col = frame[name]
            dif = self.maxs[name] - self.mins[name]
            if dif > 0:
                out, _ = K.fill_normalize(col.data, col.valid, col.fill, True, self.mins[name], dif,
                                          out_dt)
            else:
                # normalize.py:155-160: max == min -> x / (2x)  (0.5, NaN for x == 0)
                c = col.materialize()
                x = c.data.to(torch.float64)
                out = (x / (2 * x)).to(out_dt)
            new[name] = DeviceColumn(out, None, col.offsets)
        return new.to_pandas() if was_pandas else new

    def clear(self):
        self.mins = {}
        self.maxs = {}

    @property
    def output_tags(self):
        return [Tags.CONTINUOUS]

    @property
    def output_dtype(self):
        return self.out_dtype or numpy.float64
