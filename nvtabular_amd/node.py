"""Workflow graph nodes and the ``>>`` / ``+`` / ``-`` / ``[]`` DSL.

Re-creation of merlin.dag.Node (un-vendored; nvtabular/workflow/node.py:16-18
re-exports it as WorkflowNode).  Behaviour pinned by
tests/unit/workflow/test_workflow_node.py of the reference.
"""
from __future__ import annotations

from typing import List, Optional

from .schema import Schema
from .selector import ColumnSelector


class Node:
    def __init__(self, selector=None):
        self.parents: List["Node"] = []
        self.children: List["Node"] = []
        self.dependencies: List["Node"] = []
        self.op = None
        self.input_schema: Optional[Schema] = None
        self.output_schema: Optional[Schema] = None
        self._selector = None
        if selector is not None:
            self.selector = selector

    # ---- selector ----------------------------------------------------------
    @property
    def selector(self):
        return self._selector

    @selector.setter
    def selector(self, sel):
        if sel is not None and not isinstance(sel, ColumnSelector):
            sel = ColumnSelector(sel)
        self._selector = sel

    # ---- graph construction ---------------------------------------------------
    @classmethod
    def construct_from(cls, obj) -> "Node":
        if isinstance(obj, Node):
            return obj
        if isinstance(obj, (str, ColumnSelector)):
            return cls(ColumnSelector(obj) if isinstance(obj, str) else obj)
        if isinstance(obj, (list, tuple)):
            if all(isinstance(o, str) for o in obj):
                return cls(ColumnSelector(list(obj)))
            nested = [o for o in obj if isinstance(o, (list, tuple))]
            if nested and all(isinstance(o, (str, list, tuple)) for o in obj):
                return cls(ColumnSelector(list(obj)))
            nodes = [cls.construct_from(o) for o in obj]
            out = nodes[0]
            for n in nodes[1:]:
                out = out + n
            return out
        raise TypeError(f"cannot build a workflow node from {type(obj)}")

    def add_parent(self, parent):
        parent = Node.construct_from(parent)
        parent.children.append(self)
        self.parents.append(parent)

    def add_dependency(self, dep):
        dep = Node.construct_from(dep)
        dep.children.append(self)
        self.dependencies.append(dep)

    def __rshift__(self, operator):
        from .ops.base import Operator
        from .ops.lambdaop import LambdaOp

        if isinstance(operator, type) and issubclass(operator, Operator):
            operator = operator()
        elif callable(operator) and not isinstance(operator, Operator):
            operator = LambdaOp(operator)
        if not isinstance(operator, Operator):
            raise ValueError(f"Expected operator or callable, got {operator.__class__}")
        child = type(self)()
        child.op = operator
        child.add_parent(self)
        deps = operator.dependencies
        if deps is not None and deps != []:
            if isinstance(deps, (str, ColumnSelector, Node)):
                deps = [deps]
            elif isinstance(deps, (list, tuple)) and all(isinstance(d, str) for d in deps):
                deps = [list(deps)]
            for d in deps:
                child.add_dependency(d)
        return child

    def __rrshift__(self, other):
        return Node.construct_from(other) >> self

    def __add__(self, other):
        from .ops.selection import ConcatColumns

        if isinstance(self.op, ConcatColumns) and not self.children:
            child = self
        else:
            child = type(self)()
            child.op = ConcatColumns()
            child.add_parent(self)
        others = other if isinstance(other, (list, tuple)) and not all(
            isinstance(o, str) for o in other
        ) else [other]
        for o in others:
            child.add_dependency(o)
        return child

    def __radd__(self, other):
        return Node.construct_from(other) + self

    def __sub__(self, other):
        from .ops.selection import SubtractionOp

        child = type(self)()
        child.add_parent(self)
        if isinstance(other, Node):
            child.op = SubtractionOp()
            child.add_dependency(other)
        else:
            child.op = SubtractionOp(ColumnSelector(other))
        return child

    def __rsub__(self, other):
        return Node.construct_from(other) - self

    def __getitem__(self, columns):
        from .ops.selection import SubsetColumns

        child = type(self)()
        child.add_parent(self)
        child.selector = ColumnSelector(columns)
        child.op = SubsetColumns()
        return child

    # ---- schema propagation ---------------------------------------------------
    @property
    def parents_with_dependencies(self):
        return self.parents + self.dependencies

    def _upstream_selector(self, nodes) -> ColumnSelector:
        sel = ColumnSelector()
        for n in nodes:
            if n.op is None and n.selector is not None and not n.parents:
                sel = sel + n.selector.resolve(n.output_schema)
            elif n.output_schema is not None:
                part = n.output_columns
                sel = sel + part
        return sel

    def compute_schemas(self, root_schema: Schema):
        if self.op is None:
            sel = (self.selector or ColumnSelector()).resolve(root_schema)
            missing = [n for n in sel.names if n not in root_schema]
            if missing:
                raise ValueError(f"Missing columns {missing} in the dataset schema {root_schema}")
            self.input_schema = root_schema.select_by_name(sel.names)
            self.output_schema = self.input_schema
            return
        parents_schema = Schema()
        for p in self.parents:
            parents_schema = parents_schema + p.output_schema
        deps_schema = Schema()
        for d in self.dependencies:
            deps_schema = deps_schema + d.output_schema
        parents_selector = self._upstream_selector(self.parents)
        deps_selector = self._upstream_selector(self.dependencies)
        self.input_schema = self.op.compute_input_schema(
            root_schema, parents_schema, deps_schema, self.selector
        )
        self.selector = self.op.compute_selector(
            self.input_schema, self.selector, parents_selector, deps_selector
        )
        self.output_schema = self.op.compute_output_schema(self.input_schema, self.selector)

    @property
    def input_columns(self) -> ColumnSelector:
        if self.input_schema is None:
            raise RuntimeError("The input columns aren't known until the workflow is fit to a schema")
        if self.selector is not None and not self.selector.tags and not self.selector.all:
            return self.selector
        return ColumnSelector(self.input_schema.column_names)

    @property
    def output_columns(self) -> ColumnSelector:
        if self.output_schema is None:
            raise RuntimeError("The output columns aren't known until the workflow is fit to a schema")
        return ColumnSelector(self.output_schema.column_names)

    @property
    def dependency_columns(self) -> ColumnSelector:
        sel = ColumnSelector()
        for d in self.dependencies:
            sel = sel + d.output_columns
        return sel

    @property
    def label(self):
        if self.op is not None:
            return type(self.op).__name__
        return str(self.selector.names if self.selector else [])

    def __repr__(self):
        return f"<Node {self.label}>"


def iter_nodes(outputs) -> List[Node]:
    """All nodes upstream of `outputs` (inclusive), parents before children."""
    order, seen = [], set()

    def visit(n):
        if id(n) in seen:
            return
        seen.add(id(n))
        for u in n.parents_with_dependencies:
            visit(u)
        order.append(n)

    for o in outputs if isinstance(outputs, (list, tuple)) else [outputs]:
        visit(o)
    return order
