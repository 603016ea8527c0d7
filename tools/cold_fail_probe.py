"""Cold fits of the bench frame, repeated with fresh workflows: does the range path overflow on
the first partition, on which column and why (NVT_OVF_* bits)?"""
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
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 31337
dev = torch.device("cuda", 0)
frame = bench.synth_criteo(rows, dev, seed=seed)
cats = [c for c in frame.columns if c.startswith("C")]
conts = [c for c in frame.columns if c.startswith("I")]
ballast = []
for it in range(reps):
    wf = bench.build_workflow(cats, conts, tempfile.mkdtemp())
    op = [n.op for n in iter_nodes(wf.output_node) if type(n.op).__name__ == "Categorify"][0]
    K.STATS["count_relaunches"] = 0
    wf.fit(nvt.Dataset(frame))
    print(it, "failures", op._range_failures, "relaunches", K.STATS["count_relaunches"],
          "hints", {k: v for k, v in op._cap_hints.items() if "#" in k and v > 2_000_000}, flush=True)
    # change the allocator's state between repetitions (uninitialised-scratch hunting)
    ballast.append(torch.full((1 << 24,), -1 if it % 2 else 0x7F7F7F7F, dtype=torch.int32, device=dev))
    if it % 3 == 2:
        ballast.clear()
        torch.cuda.empty_cache()
