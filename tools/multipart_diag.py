"""Which counting path every column of a multi-partition cfg2 fit takes, partition by partition."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import nvtabular_amd as nvt  # noqa: E402
from nvtabular_amd import kernels as K  # noqa: E402
from nvtabular_amd.node import iter_nodes  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 45_000_000
nparts = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
frames = [bench.synth_criteo(rows, dev, seed=31337 + 1000 * p) for p in range(nparts)]
cats = [c for c in frames[0].columns if c.startswith("C")]
conts = [c for c in frames[0].columns if c.startswith("I")]
wf = bench.build_workflow(cats, conts, tempfile.mkdtemp())
op = [n.op for n in iter_nodes(wf.output_node) if type(n.op).__name__ == "Categorify"][0]
orig = op._absorb_pending


def spy(state):
    had = id(state) in op._pending_counts
    orig(state)
    if had:
        print("paths", {k: v for k, v in op._last_paths.items() if v not in (0, 6)},
              "no_range", sorted(op._no_range), "relaunches", K.STATS["count_relaunches"], flush=True)


op._absorb_pending = spy
for it in range(2):
    print("fit", it, flush=True)
    wf.fit(nvt.Dataset(frames))
print("range failures", op._range_failures)
print("hints", {k: v for k, v in op._cap_hints.items() if v > 11000})
