"""HashedCross (reference: nvtabular/ops/hashed_cross.py:25-92): XOR of the per-column hashes
of a column group, modulo ``num_buckets``, as int32 ``a_X_b``.  Same device primitive as the
combo branch of Categorify's hashed OOV buckets (``nvt_hash_bucket_*`` with an XOR carry);
hash definition in DESIGN.md section 4 (self-consistent, the reference's hash is unpinned)."""
from __future__ import annotations

from typing import Dict, Union

import numpy

from .. import kernels as K
from ..device import DeviceColumn, DeviceFrame, as_device_frame, key_view
from ..schema import Tags
from ..selector import ColumnSelector
from .base import Operator


class HashedCross(Operator):
    def __init__(self, num_buckets: Union[int, Dict[tuple, int]]):
        super().__init__()
        if not isinstance(num_buckets, (int, dict)):
            raise ValueError(f"num_buckets should be an int or dict, found {num_buckets.__class__}")
        self.num_buckets = num_buckets

    def transform(self, col_selector: ColumnSelector, df):
        frame, was_pandas = as_device_frame(df)
        out = DeviceFrame()
        for cross in _nest_columns(col_selector):
            nb = self.num_buckets[cross] if isinstance(self.num_buckets, dict) else self.num_buckets
            acc, val = None, None
            for i, column in enumerate(cross):
                keys, valid = key_view(frame[column].materialize())
                last = i == len(cross) - 1
                val, acc = K.hash_bucket(keys, nb, xor_in=acc, want_hash=not last, want_bucket=last,
                                         valid=valid)
            out["_X_".join(cross)] = DeviceColumn(val, None, None)
        return out.to_pandas() if was_pandas else out

    def column_mapping(self, col_selector):
        return {"_X_".join(cross): [*cross] for cross in _nest_columns(col_selector)}

    @property
    def output_tags(self):
        return [Tags.CATEGORICAL]

    @property
    def output_dtype(self):
        return numpy.int32


def _nest_columns(columns):
    """hashed_cross.py:84-91: a ColumnSelector is read through its flat ``.names``, so the
    whole selection is ONE cross; only a plain nested list gives several crosses."""
    if isinstance(columns, ColumnSelector):
        columns = columns.names
    if all(isinstance(col, str) for col in columns):
        return [tuple(columns)]
    return [tuple(c) for c in columns]
