"""Host logic of the drop-in boundary: the graph DSL and schema propagation
(ported from /root/reference/tests/unit/workflow/test_workflow_node.py, lines cited)."""
import os

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


# ---- graph.json persistence (reference: nvtabular/workflow/graph_serializer.py) ----------
def _named_double(col):
    return col * 2


def test_graph_json_roundtrip_layout(tmp_path):
    """Layout + vocabulary of the reference's pickle-free format (graph_serializer.py:15-29,
    985-1021): metadata.json, graph.json (format_version 1, leaves first, reference op class
    paths), artifacts/node_<id>/ holding the Categorify files; load rebuilds an equal graph."""
    import json

    import numpy as np
    import pandas as pd

    import nvtabular_amd as nvt
    from nvtabular_amd import ops
    from nvtabular_amd.schema import Schema

    stats = tmp_path / "stats" / "categories"
    stats.mkdir(parents=True)
    pd.DataFrame({"c": [5, 7], "c_size": [3, 1]}, index=[3, 4]).to_parquet(stats / "unique.c.parquet")
    pd.DataFrame({"kind": ["pad", "null", "oov", "unique"], "offset": [0, 1, 2, 3],
                  "num_indices": [1, 1, 1, 2]}).to_parquet(stats / "meta.c.parquet")
    cat = ops.Categorify(out_path=str(tmp_path / "stats"), freq_threshold=2, num_buckets={"c": 4},
                         dtype=np.int32)
    cat.categories = {"c": str(stats / "unique.c.parquet")}
    cat.storage_name = {"c": "c"}
    norm = ops.Normalize(out_dtype=np.float32)
    norm.means, norm.stds = {"x": 1.5}, {"x": 0.25}
    graph = (["c"] >> cat) + (["x"] >> ops.FillMissing(fill_val=3) >> ops.Clip(min_value=0) >> norm
                              >> ops.Rename(postfix="_n")) + (["x"] >> ops.LambdaOp(_named_double)
                                                              >> ops.Rename(name="x2")) + ["y"]
    wf = nvt.Workflow(graph)
    wf.fit_schema(Schema.from_frame(pd.DataFrame({"c": np.array([5], dtype="int32"),
                                                  "x": [1.0], "y": [0.0]})))
    out = str(tmp_path / "saved")
    wf.save(out)
    meta = json.load(open(f"{out}/metadata.json"))
    assert "nvtabular" in meta["versions"] and "generated_timestamp" in meta
    g = json.load(open(f"{out}/graph.json"))
    assert g["format_version"] == 1 and g["output_node_id"] == len(g["nodes"]) - 1
    by_cls = {}
    for rec in g["nodes"]:
        assert set(rec) == {"id", "op_class", "op_params", "op_state", "parent_ids", "dependency_ids",
                            "selector", "input_schema", "output_schema"}
        assert all(p < rec["id"] for p in rec["parent_ids"] + rec["dependency_ids"])  # leaves first
        by_cls.setdefault(rec["op_class"], []).append(rec)
    assert set(by_cls) == {
        "merlin.dag.ops.selection.SelectionOp", "merlin.dag.ops.concat_columns.ConcatColumns",
        "nvtabular.ops.categorify.Categorify", "nvtabular.ops.fill.FillMissing",
        "nvtabular.ops.clip.Clip", "nvtabular.ops.normalize.Normalize", "nvtabular.ops.rename.Rename",
        "nvtabular.ops.lambdaop.LambdaOp"}
    crec = by_cls["nvtabular.ops.categorify.Categorify"][0]
    assert crec["op_params"]["num_buckets"] == {"c": 4} and crec["op_params"]["dtype"] == {"name": "<i4"}
    assert crec["op_state"]["categories"] == [{"key": ["c"], "path": "categories/unique.c.parquet"}]
    assert os.path.exists(f"{out}/artifacts/node_{crec['id']}/categories/unique.c.parquet")
    assert os.path.exists(f"{out}/artifacts/node_{crec['id']}/categories/meta.c.parquet")
    nrec = by_cls["nvtabular.ops.normalize.Normalize"][0]
    assert nrec["op_state"] == {"means": {"x": 1.5}, "stds": {"x": 0.25}}
    assert nrec["op_params"]["out_dtype"]["name"] == "float32"
    col = crec["output_schema"][0]
    assert col["dtype"]["name"] == "int32" and col["dtype"]["element_type"] == "int"
    assert "Tags.CATEGORICAL" in col["tags"]
    lrec = by_cls["nvtabular.ops.lambdaop.LambdaOp"][0]
    assert lrec["op_params"]["f"] == {"module": _named_double.__module__, "qualname": "_named_double"}

    wf2 = nvt.Workflow.load(out)
    assert wf2.output_schema.column_names == wf.output_schema.column_names
    assert [c.dtype for c in wf2.output_schema] == [c.dtype for c in wf.output_schema]
    assert sorted(wf2.input_schema.column_names) == ["c", "x", "y"]
    ops2 = {type(n.op).__name__: n.op for n in nvt.workflow.iter_nodes(wf2.output_node) if n.op}
    assert ops2["Normalize"].means == {"x": 1.5} and ops2["Normalize"].out_dtype == np.float32
    assert ops2["Categorify"].num_buckets == {"c": 4} and ops2["Categorify"].freq_threshold == 2
    assert ops2["Categorify"].categories["c"].startswith(out) and ops2["Categorify"].dtype == np.int32
    assert ops2["FillMissing"].fill_val == 3 and ops2["Clip"].min_value == 0
    assert ops2["LambdaOp"].f is _named_double


def test_graph_json_refuses_lambdas(tmp_path):
    """graph_serializer.py:71-88."""
    import pandas as pd

    import nvtabular_amd as nvt
    from nvtabular_amd import ops
    from nvtabular_amd.graph_json import WorkflowSerializationError
    from nvtabular_amd.schema import Schema

    wf = nvt.Workflow(["x"] >> ops.LambdaOp(lambda col: col + 1))
    wf.fit_schema(Schema.from_frame(pd.DataFrame({"x": [1.0]})))
    with pytest.raises(WorkflowSerializationError):
        wf.save(str(tmp_path / "w"))


def test_groupby_schema_and_selector_cols():
    """tests/unit/ops/test_groupyby.py:131-164 of the reference: key columns appear in the output
    only when the selector names them; list aggregations are ragged lists, count is int32, sum
    float32."""
    from nvtabular_amd import Tags

    aggs = {"x": ["list", "sum"], "y": ["first", "last"], "ts": ["min", "count"]}
    full = ColumnSelector(["name", "id", "ts", "x", "y"]) >> ops.Groupby(
        groupby_cols=["name"], sort_cols=["ts"], aggs=aggs, name_sep="-")
    wf = Workflow(full).fit_schema(Schema(["name", "id", "ts", "x", "y"]))
    assert wf.output_node.output_schema.column_names == [
        "name", "x-list", "y-first", "y-last", "x-sum", "ts-min", "ts-count"]
    part = ColumnSelector(["id", "ts", "x", "y"]) >> ops.Groupby(
        groupby_cols=["name"], sort_cols=["ts"], aggs=aggs, name_sep="-")
    wf2 = Workflow(part).fit_schema(Schema(["name", "id", "ts", "x", "y"]))
    assert "name" not in wf2.output_node.output_schema.column_names
    out = wf.output_node.output_schema
    assert out["x-list"].is_list and out["x-list"].is_ragged and Tags.LIST in out["x-list"].tags
    assert out["ts-count"].dtype == np.int32 and out["x-sum"].dtype == np.float32
    assert not out["y-first"].is_list
    with pytest.raises(NotImplementedError):
        ops.Groupby(groupby_cols=["a"], aggs="median")


def test_import_nvtabular_resolves_to_this_engine():
    """Existing scripts import `nvtabular`: the alias package hands them nvtabular_amd's modules."""
    import nvtabular as nvt
    import nvtabular_amd
    from nvtabular import ops
    from nvtabular.ops import Categorify, Normalize
    from nvtabular.ops.categorify import get_embedding_sizes

    assert nvt is nvtabular_amd and ops is nvtabular_amd.ops
    assert Categorify is nvtabular_amd.ops.Categorify and Normalize is nvtabular_amd.ops.Normalize
    assert callable(get_embedding_sizes)
    graph = ["a", "b"] >> ops.Categorify() 
    assert nvt.Workflow(graph).output_node is not None and hasattr(nvt, "Dataset")
