"""Host profile (cProfile) of back-to-back cfg2 steps: where the Python time between the fit's
read-back and the transform's first launch goes."""
import cProfile
import os
import pstats
import sys
import tempfile

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
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        wf.fit(ds)
        out = wf.transform(frame)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats(os.environ.get("SORT", "cumulative")).print_stats(int(os.environ.get("TOP", 70)))
