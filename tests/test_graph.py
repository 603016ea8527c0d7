"""Host logic of the drop-in boundary: the graph DSL and schema propagation
(ported from /root/reference/tests/unit/workflow/test_workflow_node.py, lines cited)."""
import numpy as np
import pytest

import nvtabular_amd as nvt
from nvtabular_amd import ColumnSelector, Schema, Workflow, WorkflowNode, ops
from nvtabular_amd.ops import Categorify, FillMissing, Operator, Rename, TargetEncoding


# test_workflow_node.py:20-26
def test_selecting_columns_sets_selector_and_kind():
    node = ColumnSelector(["a", "b", "c"]) >> Operator()
    assert node[["a", "b"]].selector.names == ["a", "b"]
    assert node["b"].selector.names == ["b"]


# test_workflow_node.py:29-34
def test_workflow_node_converts_lists_to_selectors():
    node = WorkflowNode([])
    assert node.selector == ColumnSelector([])
    node.selector = ["a", "b", "c"]
    assert node.selector == ColumnSelector(["a", "b", "c"])


# test_workflow_node.py:37-68
def test_input_output_column_names():
    schema = Schema(["a", "b", "c", "d", "e"])
    input_node = ["a", "b", "c"] >> FillMissing()
    wf = Workflow(input_node).fit_schema(schema)
    assert wf.output_node.input_columns.names == ["a", "b", "c"]
    assert wf.output_node.output_columns.names == ["a", "b", "c"]
    wf = Workflow(input_node >> Categorify()).fit_schema(schema)
    assert wf.output_node.output_columns.names == ["a", "b", "c"]
    wf = Workflow(input_node[["b", "c"]]).fit_schema(schema)
    assert wf.output_node.input_columns.names == ["b", "c"]
    wf = Workflow(input_node + ["d"]).fit_schema(schema)
    assert wf.output_node.output_columns.names == ["a", "b", "c", "d"]
    wf = Workflow(input_node >> Rename(postfix="_renamed")).fit_schema(schema)
    assert wf.output_node.output_columns.names == ["a_renamed", "b_renamed", "c_renamed"]
    wf = Workflow(input_node >> TargetEncoding("d")).fit_schema(schema)
    assert wf.output_node.input_columns.names == ["a", "b", "c"]
    assert wf.output_node.output_columns.names == ["TE_a_d", "TE_b_d", "TE_c_d"]


# test_workflow_node.py:71-76
def test_dependency_column_names():
    dep = ["a", "b", "c"] >> TargetEncoding("d")
    Workflow(dep).fit_schema(Schema(["a", "b", "c", "d"]))
    assert dep.dependency_columns.names == ["d"]


# test_workflow_node.py:79-117
def test_workflow_node_addition():
    schema = Schema(["a", "b", "c", "d", "e", "f"])
    node1 = ["a", "b"] >> Operator()
    node2 = ["c", "d"] >> Operator()
    node3 = ["e", "f"] >> Operator()
    cols = lambda n: Workflow(n).fit_schema(schema).output_node.output_columns.names  # noqa: E731
    assert cols(node1 + node2) == ["a", "b", "c", "d"]
    assert cols(node1 + "c") == ["a", "b", "c"]
    assert cols(node1 + "c" + "d") == ["a", "b", "c", "d"]
    assert cols(node1 + node2 + "e") == ["a", "b", "c", "d", "e"]
    assert cols(node1 + node2 + node3) == ["a", "b", "c", "d", "e", "f"]
    assert cols(node1 + ["c", "d"]) == ["a", "b", "c", "d"]
    assert cols(node1 + [node2, "e"]) == ["a", "b", "c", "d", "e"]
    assert cols(node1 + [node2, node3]) == ["a", "b", "c", "d", "e", "f"]
    assert cols("c" + node1) == ["c", "a", "b"]


# test_workflow_node.py:120-156
def test_workflow_node_subtraction():
    schema = Schema(["a", "b", "c", "d", "e", "f"])
    node1 = ["a", "b", "c", "d"] >> Operator()
    node2 = ["c", "d"] >> Operator()
    cols = lambda n: Workflow(n).fit_schema(schema).output_node.output_columns.names  # noqa: E731
    assert cols(node1 - ["c", "d"]) == ["a", "b"]
    assert cols(node1 - node2) == ["a", "b"]
    assert cols(["a", "b", "c", "d"] - node2) == ["a", "b"]


# test_workflow_node.py:267-306 (nested groups for multi-column Categorify)
def test_nested_groups():
    schema = Schema(["a", "b", "c"])
    node = [["a", "b"], "c"] >> Categorify(encode_type="combo")
    wf = Workflow(node).fit_schema(schema)
    assert wf.output_node.output_columns.names == ["c", "a_b"]
    assert wf.output_node.input_columns.grouped_names == ["c", ("a", "b")]
    with pytest.raises(ValueError):
        ColumnSelector([[["a", "b"], "c"]])


def test_missing_columns_raise():
    with pytest.raises(ValueError, match="Missing columns"):
        Workflow(["a", "zzz"] >> FillMissing()).fit_schema(Schema(["a", "b"]))


def test_schema_dtypes_tags_and_properties():
    from nvtabular_amd.schema import ColumnSchema, Tags

    schema = Schema([ColumnSchema("c", np.int32), ColumnSchema("x", np.float32)])
    cats = ["c"] >> Categorify()
    conts = ["x"] >> FillMissing(add_binary_cols=True) >> ops.Normalize()
    wf = Workflow(cats + conts).fit_schema(schema)
    out = wf.output_schema
    assert out["c"].dtype == np.int64 and Tags.CATEGORICAL in out["c"].tags  # categorify.py:581-587
    assert out["x"].dtype == np.float64 and Tags.CONTINUOUS in out["x"].tags  # normalize.py:114-120
    assert out["c"].properties["embedding_sizes"] == {"cardinality": 3, "dimension": 16}
    assert out["c"].properties["domain"] == {"min": 0, "max": 2, "name": "c"}
    fm = Workflow(["x"] >> FillMissing(add_binary_cols=True)).fit_schema(schema)
    assert fm.output_schema["x_filled"].dtype == np.bool_  # fill.py:74-78
    jg = Workflow(["c"] >> ops.JoinGroupby(cont_cols=["x"], stats=["count", "mean", "sum"])).fit_schema(schema)
    assert jg.output_schema.column_names == ["c_count", "c_x_mean", "c_x_sum"]
    assert jg.output_schema["c_count"].dtype == np.int32  # join_groupby.py:29-34
    assert jg.output_schema["c_x_mean"].dtype == np.float32


def test_categorify_constructor_validation():
    # categorify.py:226-241, 287-330
    with pytest.raises(ValueError):
        Categorify(start_index=1)
    with pytest.raises(ValueError):
        Categorify(encode_type="nope")
    with pytest.raises(ValueError):
        Categorify(num_buckets=0)
    with pytest.raises(ValueError):
        Categorify(freq_threshold=2, max_size=10)
    with pytest.raises(ValueError):
        ops.JoinGroupby(cont_cols=["x"], stats=["median"])
    with pytest.raises(TypeError):
        ops.HashBucket(1.5)


def test_lambda_auto_wrap():
    # tests/unit/ops/test_lambda.py:118-120: a bare callable after >> becomes a LambdaOp
    node = ColumnSelector(["a"]) >> (lambda col: col + 1)
    assert isinstance(node.op, ops.LambdaOp)
