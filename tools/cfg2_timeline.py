"""cfg2 (headline) steps under `rocprofv3 --kernel-trace`, back to back like bench.py's timed region
(no synchronisation between steps), then 50 ms of sleep: tools/timeline_gaps.py lists the launches
of the last burst with the idle gaps in front of them."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import nvtabular_amd as nvt  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
n = int(os.environ.get("ROWS", 45_000_000))
frame = bench.synth_criteo(n, dev)
cats = [c for c in frame.columns if c.startswith("C")]
conts = [c for c in frame.columns if c.startswith("I")]
with tempfile.TemporaryDirectory() as tmp:
    wf = bench.build_workflow(cats, conts, os.path.join(tmp, "wf"))
    ds = nvt.Dataset(frame)
    for _ in range(4):
        wf.fit(ds)
        out = wf.transform(frame)
    torch.cuda.synchronize()
    for rep in range(2):
        time.sleep(0.05)
        t0 = time.perf_counter()
        for _ in range(3):
            wf.fit(ds)
            out = wf.transform(frame)
        torch.cuda.synchronize()
        print("3 steps back to back: %.3f ms per step" % (1e3 * (time.perf_counter() - t0) / 3))
