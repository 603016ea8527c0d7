"""Clip and LogOp (reference: nvtabular/ops/clip.py:25-55, logop.py:28-61) -- the default
continuous branch of the reference's Criteo benchmark
(bench/examples/dask-nvtabular-criteo-benchmark.py:201-204: FillMissing >> Clip(min_value=0)
>> LogOp).  Both are single ``nvt_clip_log`` passes and consume a pending FillMissing
constant, so that chain is two passes over the column instead of the reference's three."""
from __future__ import annotations

import numpy as np
import torch

from .. import kernels as K
from ..device import DeviceColumn, as_device_frame
from ..schema import Tags
from ..selector import ColumnSelector
from .base import Operator


class Clip(Operator):
    def __init__(self, min_value=None, max_value=None):
        if min_value is None and max_value is None:
            raise ValueError("Must specify a min or max value to clip to")
        super().__init__()
        self.min_value = min_value
        self.max_value = max_value

    def transform(self, col_selector: ColumnSelector, df):
        frame, was_pandas = as_device_frame(df)
        out = frame[col_selector.names].copy()
        for name in col_selector.names:
            col = frame[name]
            data = col.data.view(torch.uint8) if col.data.dtype == torch.bool else col.data
            out_dt = data.dtype
            bounds = [b for b in (self.min_value, self.max_value, col.fill) if b is not None]
            if out_dt in (torch.int32, torch.int64) and any(float(b) != int(b) for b in bounds):
                out_dt = torch.float64  # pandas upcasts an int column clipped to a float bound
            if out_dt == torch.uint8:
                out_dt = torch.float64
            res = K.clip_log(data, col.valid, col.fill, self.min_value, self.max_value, False, out_dt)
            keep_valid = None if (col.fill is not None or out_dt.is_floating_point) else col.valid
            out[name] = DeviceColumn(res, keep_valid, col.offsets)
        return out.to_pandas() if was_pandas else out


class LogOp(Operator):
    """log(1 + x) in float32."""

    def transform(self, col_selector: ColumnSelector, df):
        frame, was_pandas = as_device_frame(df)
        for name in col_selector.names:
            col = frame[name]
            data = col.data.view(torch.uint8) if col.data.dtype == torch.bool else col.data
            if data.dtype == torch.uint8:
                data = K.widen_i64(data)
            res = K.clip_log(data, col.valid, col.fill, None, None, True, torch.float32)
            frame[name] = DeviceColumn(res, None, col.offsets)
        return frame.to_pandas() if was_pandas else frame

    @property
    def output_tags(self):
        return [Tags.CONTINUOUS]

    @property
    def output_dtype(self):
        return np.float32
