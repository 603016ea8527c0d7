"""LambdaOp (reference: nvtabular/ops/lambdaop.py:19 -> merlin.dag.ops.udf.UDF).

The UDF is called as ``f(col)`` or ``f(col, df)`` by arity, like the reference.  On a
DeviceFrame the column handed over is a ``DeviceSeries`` (series.py): numeric expressions
(``col + 100``, ``np.log(col + 1)``, ``col.astype(float)``, ``col.fillna(0).clip(0, 5)``,
``col * df["other"]``) stay in HBM and run as device kernels.  A UDF that needs something only
pandas offers (``col.str.slice(1, 3)``) raises ``HostFallback`` inside the wrapper and is re-run
on the host with pandas Series -- the reference's CPU behaviour.  ``last_path`` records which
route the last ``transform`` took ("device" / "host").
"""
from __future__ import annotations

from inspect import signature

import pandas as pd
import torch

from ..device import DeviceColumn, DeviceFrame, as_device_frame
from ..selector import ColumnSelector
from ..series import DeviceFrameView, DeviceSeries, HostFallback
from .base import Operator


class LambdaOp(Operator):
    def __init__(self, f, dtype=None, tags=None, properties=None, dependency=None):
        super().__init__()
        if f is None:
            raise ValueError("f cannot be None. LambdaOp op applies f to dataframe")
        self.f = f
        self._param_count = len(signature(self.f).parameters)
        if self._param_count not in (1, 2):
            raise ValueError("lambda function must accept either one or two parameters")
        self.dependency = dependency
        self._dtype = dtype
        self._tags = tags or []
        self._properties = properties or {}
        self.last_path = None

    def _call(self, col, df):
        return self.f(col, df) if self._param_count == 2 else self.f(col)

    def _transform_device(self, col_selector: ColumnSelector, frame: DeviceFrame) -> DeviceFrame:
        view = DeviceFrameView(frame)
        out = DeviceFrame()
        n = len(frame)
        for name in col_selector.names:
            res = self._call(DeviceSeries.from_column(frame[name], name), view)
            if isinstance(res, DeviceSeries):
                out[name] = res.to_column()
            elif isinstance(res, torch.Tensor) and res.is_cuda and res.dim() == 1 and res.numel() == n:
                out[name] = DeviceColumn(res.contiguous())
            else:
                raise HostFallback(f"UDF returned {type(res).__name__}")
        return out

    def transform(self, col_selector: ColumnSelector, df):
        if isinstance(df, pd.DataFrame):
            host, was_pandas = df, True
        else:
            frame, _ = as_device_frame(df)
            try:
                out = self._transform_device(col_selector, frame)
                self.last_path = "device"
                return out
            except HostFallback:
                # the UDF is re-run from scratch on pandas: columns already evaluated on the
                # device are evaluated again, so a UDF with side effects sees them twice
                # (the reference calls a UDF once per column; pure functions cannot tell)
                host, was_pandas = frame.to_pandas(), False
        new = pd.DataFrame(index=host.index)
        for col in col_selector.names:
            new[col] = self._call(host[col], host)
        self.last_path = "host"
        return new if was_pandas else DeviceFrame.from_pandas(new)

    @property
    def dependencies(self):
        return self.dependency

    @property
    def dynamic_dtypes(self):
        return self._dtype is None

    @property
    def output_dtype(self):
        return self._dtype

    @property
    def output_tags(self):
        return self._tags

    @property
    def output_properties(self):
        return self._properties
