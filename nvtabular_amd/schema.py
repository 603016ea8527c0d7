"""Minimal merlin.schema re-creation: Tags, ColumnSchema, Schema.

Only what the hot-path operators read or write (categorify.py:555-587,
join_groupby.py:252-271, target_encoding.py:254-285).
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from enum import Enum
from typing import Dict, Iterable, List, Optional

import numpy as np


class Tags(Enum):
    CATEGORICAL = "categorical"
    CONTINUOUS = "continuous"
    LIST = "list"
    TARGET = "target"
    BINARY_CLASSIFICATION = "binary_classification"
    REGRESSION = "regression"
    USER = "user"
    ITEM = "item"
    ID = "id"
    EMBEDDING = "embedding"


def _norm_dtype(dt):
    if dt is None:
        return None
    try:
        import torch

        if isinstance(dt, torch.dtype):
            from .device import numpy_dtype

            return numpy_dtype(dt)
    except Exception:
        pass
    try:
        return np.dtype(dt)
    except TypeError:
        return dt


@dataclass(frozen=True)
class ColumnSchema:
    name: str
    dtype: Optional[object] = None
    tags: tuple = ()
    properties: Dict = field(default_factory=dict)
    is_list: bool = False
    is_ragged: bool = False

    def __post_init__(self):
        object.__setattr__(self, "dtype", _norm_dtype(self.dtype))
        tags = tuple(dict.fromkeys(Tags(t) if not isinstance(t, Tags) else t for t in self.tags))
        object.__setattr__(self, "tags", tags)

    def with_name(self, name):
        return replace(self, name=name)

    def with_dtype(self, dtype, is_list=None, is_ragged=None):
        return replace(
            self,
            dtype=dtype,
            is_list=self.is_list if is_list is None else is_list,
            is_ragged=self.is_ragged if is_ragged is None else is_ragged,
        )

    def with_tags(self, tags):
        if isinstance(tags, (Tags, str)):
            tags = [tags]
        return replace(self, tags=tuple(self.tags) + tuple(tags))

    def with_properties(self, props):
        return replace(self, properties={**self.properties, **props})

    def with_shape(self, is_list=False, is_ragged=False):
        return replace(self, is_list=is_list, is_ragged=is_ragged)


class Schema:
    def __init__(self, columns: Optional[Iterable] = None):
        self.column_schemas: Dict[str, ColumnSchema] = {}
        for c in columns or []:
            if isinstance(c, str):
                c = ColumnSchema(c)
            self.column_schemas[c.name] = c

    @property
    def column_names(self) -> List[str]:
        return list(self.column_schemas)

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.column_schemas[key]
        return Schema([self.column_schemas[k] for k in key])

    def get(self, name, default=None):
        return self.column_schemas.get(name, default)

    def __contains__(self, name):
        return name in self.column_schemas

    def __iter__(self):
        return iter(self.column_schemas.values())

    def __len__(self):
        return len(self.column_schemas)

    def __add__(self, other):
        if other is None:
            return self
        out = Schema(self.column_schemas.values())
        for c in other:
            out.column_schemas[c.name] = c
        return out

    def __eq__(self, other):
        return isinstance(other, Schema) and self.column_schemas == other.column_schemas

    def select_by_name(self, names) -> "Schema":
        if isinstance(names, str):
            names = [names]
        return Schema([self.column_schemas[n] for n in names if n in self.column_schemas])

    def excluding_by_name(self, names) -> "Schema":
        drop = set(names)
        return Schema([c for c in self if c.name not in drop])

    def select_by_tag(self, tags) -> "Schema":
        if isinstance(tags, (Tags, str)):
            tags = [tags]
        tags = {Tags(t) if not isinstance(t, Tags) else t for t in tags}
        return Schema([c for c in self if tags & set(c.tags)])

    def select(self, selector) -> "Schema":
        return self.select_by_name(selector.names)

    def __repr__(self):
        return f"Schema({self.column_names})"

    @staticmethod
    def from_frame(df) -> "Schema":
        """Infer a schema from a pandas DataFrame / DeviceFrame / pyarrow schema."""
        import pandas as pd

        cols = []
        if isinstance(df, pd.DataFrame):
            for n in df.columns:
                s = df[n]
                is_list = False
                dt = s.dtype
                if dt == object:
                    nn = s.dropna()
                    if len(nn) and isinstance(nn.iloc[0], (list, np.ndarray)):
                        is_list = True
                        dt = np.asarray(nn.iloc[0]).dtype if len(nn.iloc[0]) else None
                if isinstance(dt, pd.api.extensions.ExtensionDtype):
                    dt = getattr(dt, "numpy_dtype", object)
                cols.append(ColumnSchema(str(n), dt, is_list=is_list, is_ragged=is_list))
            return Schema(cols)
        from .device import DeviceFrame

        if isinstance(df, DeviceFrame):
            for n, c in df.items():
                dt = object if c.strings is not None else c.dtype
                cols.append(ColumnSchema(n, dt, is_list=c.is_list, is_ragged=c.is_list))
            return Schema(cols)
        import pyarrow as pa

        if isinstance(df, pa.Schema):
            for f in df:
                t = f.type
                is_list = pa.types.is_list(t) or pa.types.is_large_list(t)
                if is_list:
                    t = t.value_type
                try:
                    dt = t.to_pandas_dtype()
                except Exception:
                    dt = object
                cols.append(ColumnSchema(f.name, dt, is_list=is_list, is_ragged=is_list))
            return Schema(cols)
        raise TypeError(type(df))
