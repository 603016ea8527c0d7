"""Graph-plumbing operators created by the DSL (merlin.dag.ops: ConcatColumns,
SubsetColumns, SubtractionOp) plus Rename (nvtabular/ops/rename.py), which the
reference's JoinGroupby golden test chains in front of the hot path."""
from __future__ import annotations

from ..schema import Schema
from ..selector import ColumnSelector
from .base import Operator


class ConcatColumns(Operator):
    def compute_selector(self, input_schema, selector, parents_selector=None,
                         dependencies_selector=None):
        return (parents_selector or ColumnSelector()) + (dependencies_selector or ColumnSelector())

    def compute_output_schema(self, input_schema, col_selector):
        return Schema([input_schema[n] for n in col_selector.names])

    def transform(self, col_selector, df):
        return df[col_selector.names]


class SubsetColumns(Operator):
    def compute_selector(self, input_schema, selector, parents_selector=None,
                         dependencies_selector=None):
        self._validate_matching_cols(input_schema, selector, "computing input selector")
        return selector

    def compute_output_schema(self, input_schema, col_selector):
        return Schema([input_schema[n] for n in col_selector.names])

    def transform(self, col_selector, df):
        return df[col_selector.names]


class SubtractionOp(Operator):
    def __init__(self, selector=None):
        self.removed = selector

    def compute_selector(self, input_schema, selector, parents_selector=None,
                         dependencies_selector=None):
        drop = self.removed if self.removed is not None else dependencies_selector
        return (parents_selector or ColumnSelector()).filter_columns(drop or ColumnSelector())

    def compute_input_schema(self, root_schema, parents_schema, deps_schema, selector):
        return parents_schema

    def compute_output_schema(self, input_schema, col_selector):
        return Schema([input_schema[n] for n in col_selector.names])

    def transform(self, col_selector, df):
        return df[col_selector.names]


class Rename(Operator):
    """nvtabular/ops/rename.py: rename columns by function, postfix or single name."""

    def __init__(self, f=None, postfix=None, name=None):
        if not f and postfix is None and name is None:
            raise ValueError("must specify name, f, or postfix, for Rename op")
        self.f, self.postfix, self.name = f, postfix, name

    def _new_name(self, old):
        if self.f:
            return self.f(old)
        if self.postfix is not None:
            return old + self.postfix
        return self.name

    def column_mapping(self, col_selector):
        names = col_selector.names
        if self.name is not None and len(names) > 1:
            raise RuntimeError("Single column name provided for renaming multiple columns")
        return {self._new_name(n): [n] for n in names}

    def transform(self, col_selector, df):
        from ..device import as_device_frame

        frame, was_pandas = as_device_frame(df)
        mapping = {n: self._new_name(n) for n in col_selector.names}
        out = frame[col_selector.names].rename(mapping)
        return out.to_pandas() if was_pandas else out
