"""LambdaOp (reference: nvtabular/ops/lambdaop.py:19 -> merlin.dag.ops.udf.UDF).

API only: arbitrary Python cannot be a HIP kernel.  The UDF receives each column
as a pandas Series (``f(col)`` or ``f(col, df)`` by arity, like the reference's
pandas path); numeric results are moved back to HBM.  This op is outside the
measured hot path.
"""
from __future__ import annotations

from inspect import signature

import pandas as pd

from ..device import DeviceFrame, as_device_frame
from ..selector import ColumnSelector
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

    def transform(self, col_selector: ColumnSelector, df):
        frame, was_pandas = as_device_frame(df) if not isinstance(df, pd.DataFrame) else (df, True)
        host = df if isinstance(df, pd.DataFrame) else frame.to_pandas()
        new = pd.DataFrame(index=host.index)
        for col in col_selector.names:
            if self._param_count == 2:
                new[col] = self.f(host[col], host)
            else:
                new[col] = self.f(host[col])
        if was_pandas and isinstance(df, pd.DataFrame):
            return new
        return DeviceFrame.from_pandas(new)

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
