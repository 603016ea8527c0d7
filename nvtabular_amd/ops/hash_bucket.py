"""HashBucket (reference: nvtabular/ops/hash_bucket.py:28-131).

out = h32(key) % num_buckets as int32, h32 = upper half of murmur3-fmix64 of the
sign-extended key.  The reference's hash_series is
un-vendored and only pinned for range + determinism
(tests/unit/ops/test_hash_bucket.py:51-56), so bucket ids are self-consistent
across this engine (fit == transform == oracle), not reference-identical.
"""
from __future__ import annotations

from typing import Dict, Union

import numpy

from .. import kernels as K
from ..device import DeviceColumn, as_device_frame, key_view
from ..schema import Tags
from ..selector import ColumnSelector
from .base import Operator
from .categorify import _emb_sz_rule


class HashBucket(Operator):
    def __init__(self, num_buckets: Union[int, Dict[str, int]]):
        if isinstance(num_buckets, (dict, int)):
            self.num_buckets = num_buckets
        else:
            raise TypeError(
                f"`num_buckets` must be dict, iterable, or int, got type {type(num_buckets)}"
            )
        super().__init__()

    def transform(self, col_selector: ColumnSelector, df):
        frame, was_pandas = as_device_frame(df)
        if isinstance(self.num_buckets, int):
            num_buckets = {name: self.num_buckets for name in col_selector.names}
        else:
            num_buckets = self.num_buckets
        for col, nb in num_buckets.items():
            c = frame[col]
            keys, valid = key_view(c)  # nulls hash as key 0 (the reference hashes fillna(0))
            out, _ = K.hash_bucket(keys, nb, valid=valid)
            frame[col] = DeviceColumn(out, None, c.offsets)
        return frame.to_pandas() if was_pandas else frame

    def get_embedding_sizes(self, columns):
        if isinstance(self.num_buckets, int):
            return {col: _emb_sz_rule(self.num_buckets) for col in columns}
        return {col: _emb_sz_rule(self.num_buckets[col]) for col in columns}

    def _compute_properties(self, col_schema, input_schema):
        source = input_schema.column_names[0]
        cardinality, dimensions = self.get_embedding_sizes([col_schema.name])[col_schema.name]
        to_add = {}
        if cardinality and dimensions:
            to_add = {
                "domain": {"min": 0, "max": cardinality},
                "embedding_sizes": {"cardinality": cardinality, "dimension": dimensions},
            }
        return col_schema.with_properties({**input_schema[source].properties, **to_add})

    @property
    def output_tags(self):
        return [Tags.CATEGORICAL]

    @property
    def output_dtype(self):
        return numpy.int32
