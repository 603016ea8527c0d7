"""nvtabular_amd -- the NVTabular hot path (Categorify, FillMissing, Normalize,
HashBucket, JoinGroupby, TargetEncoding) as hand-written HIP kernels for MI355X
(gfx950), behind the reference's Operator / Workflow.fit + transform API.

    import nvtabular_amd as nvt
    from nvtabular_amd import ops
    cats = ["C1", "C2"] >> ops.Categorify()
    conts = ["I1"] >> ops.FillMissing() >> ops.Normalize()
    wf = nvt.Workflow(cats + conts).fit(nvt.Dataset(df))
    out = wf.transform(nvt.Dataset(df)).to_ddf().compute()

Compute has no CPU fallback: importing the package works anywhere (graph and
schema logic is host code), running an operator needs the built
``libnvt_hip.so`` and a visible GPU.
"""
from . import ops  # noqa: F401
from . import io  # noqa: F401
from .io import Dataset, Shuffle  # noqa: F401
from .node import Node  # noqa: F401
from .node import Node as WorkflowNode  # noqa: F401
from .schema import ColumnSchema, Schema, Tags  # noqa: F401
from .selector import ColumnSelector  # noqa: F401
from .workflow import Workflow  # noqa: F401

__version__ = "0.1.0"
