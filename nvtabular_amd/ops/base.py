"""Operator / StatOperator base classes.

Re-creation of merlin.dag.BaseOperator and merlin.dag.ops.stat_operator.StatOperator
(un-vendored; nvtabular/ops/operator.py:16-27 and stat_operator.py:16 re-export
them).  The method set is the drop-in boundary of SURVEY section 8(b).

Streaming fit protocol (this engine's replacement for dask task graphs): a
StatOperator implements ``fit_begin / fit_partition / fit_end``; the reference's
``fit(col_selector, ddf)`` is provided on top of it for API compatibility and
returns a lazy object with ``.compute()``.  The executor drives the streaming
protocol directly so that every StatOperator of a phase sees each partition
exactly once (one pass over the data per phase, like the fused dask graph).
"""
from __future__ import annotations

import functools
from typing import Dict, List

from ..schema import ColumnSchema, Schema
from ..selector import ColumnSelector


def _launch_locked(fn):
    """Run `fn` under kernels.LAUNCH_LOCK (re-entrant): the host-side launch sequences of two
    threads interleave at operator granularity (SURVEY 8(b) "Threading": the reference's dask
    worker threads call transform concurrently on one fitted operator)."""
    if getattr(fn, "_nvt_locked", False):
        return fn

    @functools.wraps(fn)
    def inner(self, *args, **kwargs):
        from ..kernels import LAUNCH_LOCK

        with LAUNCH_LOCK:
            return fn(self, *args, **kwargs)

    inner._nvt_locked = True
    return inner


_LOCKED_METHODS = ("transform", "fit_partition", "fit_end", "fit_finalize", "clear", "prepare_transform")


class Operator:
    """Base class for all transforms (merlin.dag.BaseOperator)."""

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        for name in _LOCKED_METHODS:
            fn = cls.__dict__.get(name)
            if callable(fn) and not isinstance(fn, (staticmethod, classmethod, property)):
                setattr(cls, name, _launch_locked(fn))

    def transform(self, col_selector: ColumnSelector, df):
        return df

    # ---- graph / schema hooks ------------------------------------------------
    @property
    def dependencies(self):
        return None

    @property
    def output_dtype(self):
        return None

    @property
    def output_tags(self):
        return []

    @property
    def output_properties(self):
        return {}

    @property
    def dynamic_dtypes(self):
        return False

    @property
    def label(self):
        return self.__class__.__name__

    @property
    def async_pending(self) -> bool:
        """True while results of this operator's fit are still being produced on the library's
        internal streams (Categorify): the executor schedules independent branches first."""
        return False

    def range_name(self, kind: str) -> str:
        """roctx range name = the reference's @annotate string for this operator
        (e.g. normalize.py:61,70 "Normalize_fit" / "Normalize_op")."""
        return f"{self.label}_{'fit' if kind == 'fit' else 'op'}"

    def column_mapping(self, col_selector: ColumnSelector) -> Dict[str, List[str]]:
        return {name: [name] for name in col_selector.names}

    def compute_selector(self, input_schema, selector, parents_selector=None,
                         dependencies_selector=None) -> ColumnSelector:
        sel = selector if selector else (parents_selector or ColumnSelector())
        self._validate_matching_cols(input_schema, sel, "computing input selector")
        return sel

    def compute_input_schema(self, root_schema, parents_schema, deps_schema, selector) -> Schema:
        return parents_schema + deps_schema

    def _validate_matching_cols(self, schema, selector, method_name):
        missing = [n for n in selector.names if n not in schema]
        if missing:
            raise ValueError(
                f"Missing columns {missing} found in operator {self.__class__.__name__} "
                f"during {method_name}."
            )

    def compute_output_schema(self, input_schema: Schema, col_selector: ColumnSelector) -> Schema:
        out = []
        for out_name, in_names in self.column_mapping(col_selector).items():
            col = ColumnSchema(out_name)
            sub = input_schema.select_by_name(in_names)
            col = self._compute_dtype(col, sub)
            col = self._compute_tags(col, sub)
            col = self._compute_properties(col, sub)
            col = self._compute_shape(col, sub)
            out.append(col)
        return Schema(out)

    def _compute_dtype(self, col_schema, input_schema):
        dtype, is_list, is_ragged = col_schema.dtype, col_schema.is_list, col_schema.is_ragged
        if input_schema.column_schemas:
            src = input_schema[input_schema.column_names[0]]
            dtype, is_list, is_ragged = src.dtype, src.is_list, src.is_ragged
        if self.output_dtype is not None:
            dtype = self.output_dtype
        return col_schema.with_dtype(dtype, is_list=is_list, is_ragged=is_ragged)

    def _compute_tags(self, col_schema, input_schema):
        tags = []
        if input_schema.column_schemas:
            tags = list(input_schema[input_schema.column_names[0]].tags)
        return col_schema.with_tags(tags + list(self.output_tags))

    def _compute_properties(self, col_schema, input_schema):
        props = {}
        if input_schema.column_schemas:
            props = dict(input_schema[input_schema.column_names[0]].properties)
        return col_schema.with_properties({**props, **self.output_properties})

    def _compute_shape(self, col_schema, input_schema):
        return col_schema

    def inference_initialize(self, col_selector, model_config):
        return None

    def __rrshift__(self, other):
        from ..node import Node

        return Node.construct_from(other) >> self


class Lazy:
    """Stand-in for dask.delayed.Delayed: ``.compute()`` runs the deferred fit."""

    def __init__(self, fn):
        self._fn = fn

    def compute(self, **_):
        return self._fn()


class StatOperator(Operator):
    """Operator with a statistics-gathering phase (merlin StatOperator)."""

    # -- streaming protocol (implemented by subclasses) --
    def fit_begin(self, col_selector: ColumnSelector):
        raise NotImplementedError

    def fit_partition(self, state, col_selector: ColumnSelector, df):
        raise NotImplementedError

    def fit_end(self, state, col_selector: ColumnSelector):
        raise NotImplementedError

    # -- reference API --
    def fit(self, col_selector: ColumnSelector, ddf):
        """``ddf`` is any iterable of partitions (DeviceFrame / pandas).  Returns a
        lazy result to hand to ``fit_finalize`` (categorify.py:346, normalize.py:62)."""

        def run():
            from ..device import as_device_frame

            state = self.fit_begin(col_selector)
            parts = ddf.to_iter() if hasattr(ddf, "to_iter") else ddf
            for part in parts:
                frame, _ = as_device_frame(part)
                self.fit_partition(state, col_selector, frame)
            return self.fit_end(state, col_selector)

        return Lazy(run)

    def fit_finalize(self, stats):
        raise NotImplementedError

    def clear(self):
        raise NotImplementedError

    def set_storage_path(self, new_path, copy=False):
        pass
