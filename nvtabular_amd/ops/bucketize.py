"""Bucketize (reference: nvtabular/ops/bucketize.py:25-110): continuous -> bin index,
``np.digitize(x, boundaries, right=False)`` as int32, one ``nvt_bucketize`` pass per column."""
from __future__ import annotations

import numpy as np
import torch

from .. import kernels as K
from ..device import DeviceColumn, as_device_frame
from ..schema import Tags
from ..selector import ColumnSelector
from .base import Operator


class Bucketize(Operator):
    def __init__(self, boundaries):
        # kept for JSON serialisation (graph_serializer.py:434-449); callables are not JSON-safe
        self._original_boundaries = boundaries if isinstance(boundaries, (list, tuple, dict)) else None
        if isinstance(boundaries, (list, tuple)):
            self.boundaries = lambda col: boundaries
        elif isinstance(boundaries, dict):
            self.boundaries = lambda col: boundaries[col]
        elif callable(boundaries):
            self.boundaries = boundaries
        else:
            raise TypeError(
                "`boundaries` must be dict, callable, or list, got type {}".format(type(boundaries))
            )
        super().__init__()

    def transform(self, col_selector: ColumnSelector, df):
        frame, was_pandas = as_device_frame(df)
        out = frame[col_selector.names].copy()
        for name in col_selector.names:
            col = frame[name].materialize()
            b = np.asarray(self.boundaries(name), dtype=np.float64)
            if b.size > 1 and not np.all(np.diff(b) >= 0):
                raise ValueError(f"Bucketize boundaries of '{name}' must be ascending")
            data = col.data.view(torch.uint8) if col.data.dtype == torch.bool else col.data
            if data.dtype == torch.uint8:
                data = K.widen_i64(data)
            bt = torch.from_numpy(b).to(data.device)
            out[name] = DeviceColumn(K.bucketize(data, col.valid, bt), None, col.offsets)
        return out.to_pandas() if was_pandas else out

    @property
    def output_tags(self):
        return [Tags.CATEGORICAL]

    @property
    def output_dtype(self):
        return np.int32
