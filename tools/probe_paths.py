"""Per-column counting path / distinct keys / max count of one cold + one steady fit of the
bench frame, and the per-scope kernel times of the steady step (families alone)."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import nvtabular_amd as nvt  # noqa: E402
from nvtabular_amd import kernels as K  # noqa: E402
from nvtabular_amd.node import iter_nodes  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 45_000_000
dev = torch.device("cuda", 0)
frame = bench.synth_criteo(rows, dev)
cats = [f"C{i + 1}" for i in range(26)]
conts = [f"I{i + 1}" for i in range(13)]
tmp = tempfile.mkdtemp()
wf = bench.build_workflow(cats, conts, tmp)
ds = nvt.Dataset(frame)
for it in range(3):
    wf.fit(ds)
    out = wf.transform(frame)
    del out
torch.cuda.synchronize()
op = [n.op for n in iter_nodes(wf.output_node) if type(n.op).__name__ == "Categorify"][0]
for c in cats:
    print(c, "path", op._last_paths.get(f"{c}#0"), "distinct", op._cap_hints.get(f"{c}#0"),
          "no_range" if f"{c}#0" in op._no_range else "")
K.profile_begin()
wf.fit(ds)
out = wf.transform(frame)
rep = K.profile_report()
for k, v in sorted(rep["kernels"].items()):
    print(f"{k:24s} {v[0]:8.3f} ms  {v[1]:4d} launches")
print("busy", rep["busy_ms"])
