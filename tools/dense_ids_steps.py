"""The dense-ids frame (bench.py cfg2_dense_ids) step by step: wall time, counting paths and
relaunches of every step -- hunting the step that takes hundreds of ms."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import nvtabular_amd as nvt  # noqa: E402
from nvtabular_amd import kernels as K  # noqa: E402
from nvtabular_amd.node import iter_nodes  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
frame = bench.synth_criteo(45_000_000, dev, scramble=False)
cats = [c for c in frame.columns if c.startswith("C")]
conts = [c for c in frame.columns if c.startswith("I")]
wf = bench.build_workflow(cats, conts, os.path.join(tempfile.mkdtemp(prefix="nvt_di_"), "wf"))
ds = nvt.Dataset(frame)
op = [n.op for n in iter_nodes(wf.output_node) if type(n.op).__name__ == "Categorify"][0]
for i in range(int(os.environ.get("STEPS", 14))):
    K.STATS["count_relaunches"] = 0
    a0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wf.fit(ds)
    t1 = time.perf_counter()
    out = wf.transform(frame)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    paths = {}
    for h, v in op._last_paths.items():
        paths.setdefault(v, []).append(h)
    odd = {p: hs for p, hs in paths.items() if p not in (0, 6, 9)}
    print("step %2d  fit %8.2f ms  transform %8.2f ms  relaunches %d  allocs %d  other paths %s  pieces %s  failures %s" % (
        i, 1e3 * (t1 - t0), 1e3 * (t2 - t1), K.STATS["count_relaunches"],
        torch.cuda.memory_stats().get("num_device_alloc", 0) - a0, odd, sorted(op._range_pieces),
        [(h, b) for h, b, _ in op._range_failures]), flush=True)
