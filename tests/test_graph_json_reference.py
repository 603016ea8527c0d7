"""Mechanical cross-check of graph.json against the REFERENCE serializer's source
(nvtabular/workflow/graph_serializer.py:130-300 helpers, :295-820 per-operator (to_dict, from_dict)
pairs, :830-896 registry, :985-1075 node records, :1077-1165 graph file).

The reference cannot be imported here (merlin-core is absent), but its source can be parsed: for every
operator class this engine serialises,
  * every key the reference's `_<op>_from_dict` READS from `params` / `state` must be among the keys
    this engine WRITES for that `op_class` (a stock NVTabular can open our file), and
  * every key the reference's `_<op>_to_dict` WRITES must be among the keys this engine's
    deserializer READS or deliberately ignores (we can open a stock NVTabular file),
and the same for the node records, the graph file, column schemas, dtypes, selectors and the
{"key", "path"} artifact records.  Skipped where /root/reference does not exist (the GPU box)."""
import ast
import json
import os

import numpy as np
import pandas as pd
import pytest

REF = "/root/reference/nvtabular/workflow/graph_serializer.py"
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="reference source not present")


# ---- what the reference's source says ------------------------------------------------------------
def _const_str(node):
    return node.value if isinstance(node, ast.Constant) and isinstance(node.value, str) else None


def _keys_read(fn, names):
    """{name: keys} read through name.get("k"), name["k"], "k" in name -- also through a loop
    variable or comprehension variable that iterates over `name` (attributed to name + "[]")."""
    out = {n: set() for n in names}
    alias = {n: n for n in names}
    for node in ast.walk(fn):
        it = None
        if isinstance(node, ast.For):
            it = (node.target, node.iter)
        if isinstance(node, ast.comprehension):
            it = (node.target, node.iter)
        if it and isinstance(it[0], ast.Name):
            src = it[1]
            if isinstance(src, ast.Subscript) and isinstance(src.value, ast.Name) and src.value.id in alias:
                k = _const_str(src.slice)   # for dim in d["shape"]
                if k:
                    alias[it[0].id] = alias[src.value.id] + "." + k + "[]"
                    out.setdefault(alias[it[0].id], set())
            elif isinstance(src, ast.Name) and src.id in alias:
                alias[it[0].id] = alias[src.id] + "[]"
                out.setdefault(alias[it[0].id], set())
    for node in ast.walk(fn):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "get" \
                and isinstance(node.func.value, ast.Name) and node.func.value.id in alias and node.args:
            k = _const_str(node.args[0])
            if k:
                out[alias[node.func.value.id]].add(k)
        if isinstance(node, ast.Subscript) and isinstance(node.value, ast.Name) and node.value.id in alias \
                and isinstance(node.ctx, ast.Load):
            k = _const_str(node.slice)
            if k:
                out[alias[node.value.id]].add(k)
        if isinstance(node, ast.Compare) and len(node.ops) == 1 and isinstance(node.ops[0], ast.In) \
                and isinstance(node.comparators[0], ast.Name) and node.comparators[0].id in alias:
            k = _const_str(node.left)
            if k:
                out[alias[node.comparators[0].id]].add(k)
    return out


def _dict_keys(node, env):
    """Top-level string keys of a dict expression (a literal, or a name bound to one in `env`,
    plus name["k"] = ... stores)."""
    if isinstance(node, ast.Dict):
        return {k.value for k in node.keys if isinstance(k, ast.Constant) and isinstance(k.value, str)}
    if isinstance(node, ast.Name):
        return set(env.get(node.id, set()))
    return set()


def _keys_written(fn):
    """(params keys, state keys) of an `_<op>_to_dict`: its `return (params, state)`."""
    env = {}
    for node in ast.walk(fn):
        if isinstance(node, ast.Assign) and len(node.targets) == 1:
            t = node.targets[0]
            if isinstance(t, ast.Name) and isinstance(node.value, ast.Dict):
                env.setdefault(t.id, set()).update(_dict_keys(node.value, env))
            if isinstance(t, ast.Subscript) and isinstance(t.value, ast.Name) and _const_str(t.slice):
                env.setdefault(t.value.id, set()).add(_const_str(t.slice))
    params, state = set(), set()
    for node in ast.walk(fn):
        if isinstance(node, ast.Return) and isinstance(node.value, ast.Tuple) and len(node.value.elts) == 2:
            params |= _dict_keys(node.value.elts[0], env)
            state |= _dict_keys(node.value.elts[1], env)
    return params, state


def _single_dict_written(fn):
    env, keys = {}, set()
    for node in ast.walk(fn):
        if isinstance(node, ast.Assign) and len(node.targets) == 1:
            t = node.targets[0]
            if isinstance(t, ast.Name) and isinstance(node.value, ast.Dict):
                env.setdefault(t.id, set()).update(_dict_keys(node.value, env))
            if isinstance(t, ast.Subscript) and isinstance(t.value, ast.Name) and _const_str(t.slice):
                env.setdefault(t.value.id, set()).add(_const_str(t.slice))
    for node in ast.walk(fn):
        if isinstance(node, ast.Return) and node.value is not None:
            keys |= _dict_keys(node.value, env)
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "append" \
                and node.args and isinstance(node.args[0], ast.Dict):
            keys |= _dict_keys(node.args[0], env)
    return keys


@pytest.fixture(scope="module")
def ref():
    tree = ast.parse(open(REF).read())
    fns = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
    registry = {}   # class path -> (to_dict name, from_dict name)
    for node in ast.walk(fns["_build_registry"]):
        if isinstance(node, ast.Tuple) and len(node.elts) == 4 and _const_str(node.elts[0]) \
                and all(isinstance(e, ast.Name) for e in node.elts[2:]):
            registry[_const_str(node.elts[0])] = (node.elts[2].id, node.elts[3].id)
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id == "_register" \
                and len(node.args) == 4 and _const_str(node.args[0]) and all(isinstance(a, ast.Name) for a in node.args[2:]):
            registry[_const_str(node.args[0])] = (node.args[2].id, node.args[3].id)
    assert len(registry) >= 19, registry
    return fns, registry


# ---- what this engine writes and reads -------------------------------------------------------------
class _Recording(dict):
    """dict that remembers which keys were asked for."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.asked = set()

    def get(self, key, default=None):
        self.asked.add(key)
        return super().get(key, default)

    def __getitem__(self, key):
        self.asked.add(key)
        return super().__getitem__(key)

    def __contains__(self, key):
        self.asked.add(key)
        return super().__contains__(key)


def _double(col):
    return col * 2


@pytest.fixture(scope="module")
def ours(tmp_path_factory):
    """A saved workflow holding every operator this engine serialises -> graph.json records by
    op_class, and the keys our deserializer asked for per op_class."""
    import nvtabular_amd as nvt
    from nvtabular_amd import graph_json, ops
    from nvtabular_amd.schema import Schema

    tmp = tmp_path_factory.mktemp("gj")
    stats = tmp / "stats" / "categories"
    stats.mkdir(parents=True)
    pd.DataFrame({"c": [5, 7], "c_size": [3, 1]}, index=[3, 4]).to_parquet(stats / "unique.c.parquet")
    pd.DataFrame({"kind": ["pad", "null", "oov", "unique"], "offset": [0, 1, 2, 3],
                  "num_indices": [1, 1, 1, 2]}).to_parquet(stats / "meta.c.parquet")
    cat = ops.Categorify(out_path=str(tmp / "stats"), freq_threshold=2, num_buckets={"c": 4}, dtype=np.int32)
    cat.categories = {"c": str(stats / "unique.c.parquet")}
    cat.storage_name = {"c": "c"}
    jdir = tmp / "jg" / "categories"
    jdir.mkdir(parents=True)
    (jdir / "cat_stats.k.parquet").mkdir()   # (a directory of parts, like the reference's dask output)
    pd.DataFrame({"k": [1, 2], "k_count": [2, 1], "k_x_sum": [1.0, 2.0]}).to_parquet(
        jdir / "cat_stats.k.parquet" / "part.0.parquet")
    jg = ops.JoinGroupby(cont_cols=["x"], stats=["count", "sum"], out_path=str(tmp / "jg"))
    jg.categories = {"k": str(jdir / "cat_stats.k.parquet")}
    jg.storage_name = {"k": "k"}
    tdir = tmp / "te" / "categories"
    tdir.mkdir(parents=True)
    (tdir / "cat_stats.k.parquet").mkdir()   # (TargetEncoding's statistics are a directory of parts)
    pd.DataFrame({"k": [1, 2], "__fold__": [0, 1], "y_sum": [1.0, 0.0], "y_count": [2, 1]}).to_parquet(
        tdir / "cat_stats.k.parquet" / "part.0.parquet")
    te = ops.TargetEncoding("y", kfold=2, out_path=str(tmp / "te"))
    te.stats = {"k": str(tdir / "cat_stats.k.parquet")}
    te.means = {"y": 0.5}
    norm = ops.Normalize(out_dtype=np.float32)
    norm.means, norm.stds = {"x": 1.5}, {"x": 0.25}
    mm = ops.NormalizeMinMax()
    mm.mins, mm.maxs = {"x": 0.0}, {"x": 2.0}
    graph = ((["c"] >> cat) + (["x"] >> ops.FillMissing(fill_val=3) >> ops.Clip(min_value=0) >> norm
                               >> ops.Rename(postfix="_n"))
             + (["x"] >> ops.LogOp() >> mm >> ops.Rename(name="x_mm"))
             + (["x"] >> ops.LambdaOp(_double) >> ops.Rename(name="x2"))
             + (["x"] >> ops.Bucketize({"x": [0.5, 1.5]}) >> ops.Rename(name="xb"))
             + (["k"] >> ops.HashBucket(16) >> ops.Rename(name="kh"))
             + (["k"] >> jg) + (["k"] >> te) + ["y"])
    wf = nvt.Workflow(graph)
    wf.fit_schema(Schema.from_frame(pd.DataFrame({"c": np.array([5], dtype="int32"), "x": [1.0],
                                                  "k": np.array([1], dtype="int32"),
                                                  "y": np.array([0.0], dtype="float32")})))
    out = str(tmp / "saved")
    wf.save(out)
    g = json.load(open(os.path.join(out, "graph.json")))
    by_cls = {}
    for rec in g["nodes"]:
        by_cls.setdefault(rec["op_class"], []).append(rec)
    # load it again through dicts that record what the deserializer asks for
    asked = {}
    real_load = graph_json.json.load

    def wrap(o):
        if isinstance(o, dict):
            return _Recording({k: wrap(v) for k, v in o.items()})
        if isinstance(o, list):
            return [wrap(v) for v in o]
        return o

    def load(f):
        doc = wrap(real_load(f))
        asked["doc"] = doc
        return doc

    graph_json.json.load = load
    try:
        nvt.Workflow.load(out)
    finally:
        graph_json.json.load = real_load
    return g, by_cls, asked["doc"]


def _ours_asked(doc, op_class):
    params, state = set(), set()
    for rec in doc["nodes"]:
        if dict.get(rec, "op_class") == op_class:
            params |= dict.get(rec, "op_params").asked
            state |= dict.get(rec, "op_state").asked
    return params, state


# the reference's serializer covers operators outside this engine's scope (not on
# the hot path) -- they are named here so that a new one shows up as a failure, not silently
_REFERENCE_ONLY = {"nvtabular.ops.fill.FillMedian", "nvtabular.ops.list_slice.ListSlice", "nvtabular.ops.dropna.Dropna",
                   "nvtabular.ops.add_metadata.AddMetadata", "nvtabular.ops.filter.Filter",
                   "merlin.dag.ops.subgraph.Subgraph"}


def test_registry_class_paths(ref, ours):
    _, registry = ref
    from nvtabular_amd import graph_json

    mine = set(graph_json._registry()) | {graph_json._SELECTION}
    assert mine <= set(registry), mine - set(registry)          # no class path the reference does not know
    assert set(registry) - mine == _REFERENCE_ONLY, set(registry) - mine - _REFERENCE_ONLY


def test_every_key_the_reference_reads_is_written(ref, ours):
    fns, registry = ref
    g, by_cls, _ = ours
    checked = 0
    for cls, recs in by_cls.items():
        _, from_name = registry[cls]
        reads = _keys_read(fns[from_name], ("params", "state"))
        for rec in recs:
            missing_p = reads["params"] - set(rec["op_params"])
            missing_s = reads["state"] - set(rec["op_state"])
            assert not missing_p and not missing_s, (cls, "the reference reads", missing_p, missing_s)
            checked += 1
    assert checked >= 14 and {"nvtabular.ops.categorify.Categorify", "nvtabular.ops.join_groupby.JoinGroupby",
                              "nvtabular.ops.target_encoding.TargetEncoding"} <= set(by_cls)


def test_every_key_the_reference_writes_is_read(ref, ours):
    fns, registry = ref
    g, by_cls, doc = ours
    for cls in by_cls:
        to_name, _ = registry[cls]
        wp, ws = _keys_written(fns[to_name])
        ap, as_ = _ours_asked(doc, cls)
        # written by BOTH sides and not needed to rebuild the operator here
        ignorable = {"merlin.dag.ops.selection.SelectionOp": {"selector"}}.get(cls, set())
        assert wp - ap - ignorable == set(), (cls, "params written by the reference, never read here", wp - ap)
        assert ws - as_ == set(), (cls, "state written by the reference, never read here", ws - as_)
        # ... and the reference's writer emits nothing our writer does not
        for rec in by_cls[cls]:
            assert wp <= set(rec["op_params"]) and ws <= set(rec["op_state"]), (cls, wp, ws, rec)


def test_node_records_and_graph_file(ref, ours):
    fns, _ = ref
    g, by_cls, doc = ours
    node_reads = _keys_read(fns["_deserialize_node"], ("record",))["record"]
    node_writes = _single_dict_written(fns["_serialize_node"])
    assert node_reads and node_reads <= node_writes
    for rec in g["nodes"]:
        assert set(rec) == node_writes, (set(rec) ^ node_writes)
    graph_reads = set()
    for name in ("deserialize_graph",):
        for var in ("graph", "data", "doc", "d", "graph_dict", "graph_data"):
            graph_reads |= _keys_read(fns[name], (var,))[var]
    graph_writes = set()
    for node in ast.walk(fns["serialize_graph"]):
        if isinstance(node, ast.Dict):
            ks = {k.value for k in node.keys if isinstance(k, ast.Constant) and isinstance(k.value, str)}
            if "nodes" in ks:
                graph_writes = ks
    assert graph_writes and graph_reads <= graph_writes, (graph_reads, graph_writes)
    assert set(g) == graph_writes, set(g) ^ graph_writes
    assert {"nodes", "output_node_id"} <= doc.asked


def test_schema_dtype_selector_and_artifact_records(ref, ours):
    fns, _ = ref
    g, by_cls, doc = ours
    cs_reads = _keys_read(fns["_column_schema_from_dict"], ("d",))["d"]
    dt_reads = _keys_read(fns["_dtype_from_dict"], ("d",))
    sel_reads = _keys_read(fns["_selector_from_dict"], ("d",))["d"]
    art_reads = _keys_read(fns["_categories_from_json"], ("records",))["records[]"]
    assert cs_reads == {"name", "tags", "properties", "dtype", "is_list", "is_ragged"}
    assert art_reads == {"key", "path"} and sel_reads == {"names"}
    optional_dtype = {"element_type", "element_unit", "shape", "element_size", "signed"}   # `in d` / .get
    seen_dtype_keys = set()
    for rec in g["nodes"]:
        for schema in (rec["input_schema"], rec["output_schema"]):
            for col in schema or []:
                assert cs_reads <= set(col), (rec["op_class"], cs_reads - set(col))
                assert dt_reads["d"] - optional_dtype <= set(col["dtype"]), col["dtype"]
                seen_dtype_keys |= set(col["dtype"])
                for dim in col["dtype"].get("shape", []):
                    assert dt_reads.get("d.shape[]", set()) <= set(dim)
        if rec["selector"] is not None:
            assert sel_reads <= set(rec["selector"])
    assert {"name", "element_type", "element_size"} <= seen_dtype_keys
    for cls, key in (("nvtabular.ops.categorify.Categorify", "categories"),
                     ("nvtabular.ops.join_groupby.JoinGroupby", "categories"),
                     ("nvtabular.ops.target_encoding.TargetEncoding", "stats")):
        recs = by_cls[cls][0]["op_state"][key]
        assert recs and all(set(r) == art_reads for r in recs), (cls, recs)
        assert not any(os.path.isabs(r["path"]) for r in recs)   # relative to artifacts/node_<id>/
