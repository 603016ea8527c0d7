"""``import nvtabular`` resolves to this engine: existing NVTabular scripts (``import nvtabular as
nvt``, ``from nvtabular import ops``, ``from nvtabular.ops import Categorify``, ``nvt.Workflow``,
``nvt.Dataset``) run on the MI355X kernels of ``nvtabular_amd`` without an edit.  Nothing lives
here: the module objects of ``nvtabular_amd`` are registered under the reference's names."""
import importlib
import sys

import nvtabular_amd as _impl

for _name in ("ops", "workflow", "io", "schema", "selector", "node", "graph_json", "dist"):
    sys.modules[f"{__name__}.{_name}"] = importlib.import_module(f"nvtabular_amd.{_name}")
for _sub in ("categorify", "normalize", "fill", "join_groupby", "target_encoding", "hash_bucket",
             "lambdaop", "groupby", "clip_log", "hashed_cross", "bucketize"):
    try:
        sys.modules[f"{__name__}.ops.{_sub}"] = importlib.import_module(f"nvtabular_amd.ops.{_sub}")
    except ImportError:  # (an operator module this engine does not carry)
        pass
sys.modules[__name__] = _impl
