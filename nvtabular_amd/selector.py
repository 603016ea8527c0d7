"""ColumnSelector -- re-creation of merlin.dag.ColumnSelector (un-vendored
dependency of nvtabular; semantics pinned by the call sites in
nvtabular/ops/categorify.py:350-365,385 and tests/unit/workflow/test_workflow_node.py)."""
from __future__ import annotations

from typing import List, Union


class ColumnSelector:
    """A list of column names, optionally with nested groups (``[["a","b"], "c"]``)
    that multi-column operators treat as one unit."""

    def __init__(self, names=None, subgroups=None, tags=None):
        self._names: List[str] = []
        self.subgroups: List["ColumnSelector"] = list(subgroups or [])
        self.tags = list(tags or [])
        self.all = names == "*"
        if names is None or self.all:
            names = []
        if isinstance(names, ColumnSelector):
            self._names = list(names._names)
            self.subgroups = list(names.subgroups) + self.subgroups
            self.tags = list(names.tags) + self.tags
            return
        if isinstance(names, str):
            names = [names]
        for n in names:
            if isinstance(n, str):
                self._names.append(n)
            elif isinstance(n, ColumnSelector):
                self.subgroups.append(n)
            elif isinstance(n, (list, tuple)):
                for sub in n:
                    if not isinstance(sub, str):
                        raise ValueError("Too many nested levels in column groups")
                self.subgroups.append(ColumnSelector(list(n)))
            else:
                raise TypeError(f"bad column selector entry {n!r}")

    @property
    def names(self) -> List[str]:
        out = list(self._names)
        for g in self.subgroups:
            for n in g.names:
                if n not in out:
                    out.append(n)
        return out

    @property
    def grouped_names(self) -> List[Union[str, tuple]]:
        out: list = list(self._names)
        for g in self.subgroups:
            out.append(tuple(g.names))
        return out

    def resolve(self, schema) -> "ColumnSelector":
        if self.all:
            return ColumnSelector(list(schema.column_names))
        names = list(self._names)
        if self.tags:
            names += [n for n in schema.select_by_tag(self.tags).column_names if n not in names]
        return ColumnSelector(names, subgroups=self.subgroups)

    def __add__(self, other):
        if other is None:
            return self
        if isinstance(other, (str, list, tuple)):
            other = ColumnSelector(other)
        if not isinstance(other, ColumnSelector):
            return NotImplemented
        names = list(self._names) + [n for n in other._names if n not in self._names]
        return ColumnSelector(names, subgroups=self.subgroups + other.subgroups,
                              tags=self.tags + other.tags)

    __radd__ = __add__

    def __rshift__(self, op):
        from .node import Node

        return Node(self) >> op

    def __eq__(self, other):
        if not isinstance(other, ColumnSelector):
            return False
        return (self._names == other._names and self.subgroups == other.subgroups
                and self.all == other.all)

    def __bool__(self):
        return bool(self.all or self._names or self.subgroups or self.tags)

    def __len__(self):
        return len(self.names)

    def __repr__(self):
        return f"ColumnSelector({self.grouped_names!r})"

    def filter_columns(self, other: "ColumnSelector") -> "ColumnSelector":
        drop = set(other.names)
        kept_groups = [g for g in self.subgroups if not (set(g.names) & drop)]
        return ColumnSelector([n for n in self._names if n not in drop], subgroups=kept_groups)
