"""FillMissing (reference: nvtabular/ops/fill.py:25-80).

On a DeviceFrame the fill is *deferred*: the column records the constant and
every downstream kernel (moments, normalize, encode) takes it as a parameter, so
FillMissing >> Normalize is one pass over the data instead of the reference's
two (fill.py:55 then normalize.py:80).  The fill is materialised (one
``nvt_fill_normalize`` pass, do_norm=0) only if the filled column itself is output.
"""
from __future__ import annotations

import torch

from .. import kernels as K
from ..device import DeviceColumn, as_device_frame
from ..selector import ColumnSelector
from .base import Operator


class FillMissing(Operator):
    def __init__(self, fill_val=0, add_binary_cols=False):
        super().__init__()
        self.fill_val = fill_val
        self.add_binary_cols = add_binary_cols

    def transform(self, col_selector: ColumnSelector, df):
        frame, was_pandas = as_device_frame(df)
        for name in col_selector.names:
            col = frame[name]
            if self.add_binary_cols:
                data = col.data.view(torch.uint8) if col.data.dtype == torch.bool else col.data
                out_dt = data.dtype if data.dtype != torch.uint8 else torch.float64
                fv = float(self.fill_val)
                if out_dt in (torch.int32, torch.int64) and fv != int(fv):
                    out_dt = torch.float64
                out, filled = K.fill_normalize(data, col.valid, fv, False, 0.0, 1.0, out_dt,
                                               want_filled_mask=True)
                frame[f"{name}_filled"] = DeviceColumn(filled)
                frame[name] = DeviceColumn(out, None, col.offsets)
            else:
                pending = col.shallow_copy()
                if pending.fill is None:  # an earlier fill already decided null rows
                    pending.fill = self.fill_val
                frame[name] = pending
        return frame.to_pandas() if was_pandas else frame

    def column_mapping(self, col_selector):
        mapping = super().column_mapping(col_selector)
        for name in col_selector.names:
            if self.add_binary_cols:
                mapping[f"{name}_filled"] = [name]
        return mapping

    def _compute_dtype(self, col_schema, input_schema):
        col_schema = super()._compute_dtype(col_schema, input_schema)
        if col_schema.name.endswith("_filled"):
            col_schema = col_schema.with_dtype(bool)
        return col_schema
