"""Where the host spends the time between the fit's last read-back (the count states) and the
transform's first launch (fill + normalize) in back-to-back cfg2 steps: wall-clock marks around
the functions on that path (no profiler: its overhead is larger than what is measured)."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import nvtabular_amd as nvt  # noqa: E402
from nvtabular_amd import kernels as K  # noqa: E402
from nvtabular_amd.ops import categorify as CAT  # noqa: E402

marks = []


def wrap(obj, name, tag):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            marks.append((tag, t0, time.perf_counter()))
    setattr(obj, name, g)


wrap(K.CountBatch, "_read_states", "read_states")
wrap(K.DenseCountJob, "resolve", "resolve (x26)")
wrap(CAT.Categorify, "_merge_parts_many", "merge_parts_many")
wrap(CAT.Categorify, "_complete_sorted_info", "complete_sorted_info")
from nvtabular_amd.ops import normalize as NORM  # noqa: E402
wrap(NORM.Normalize, "fit_end", "normalize_fit_end")
wrap(NORM.Normalize, "transform", "normalize_transform")
wrap(nvt.Workflow, "_run", "workflow_run")
wrap(CAT.Categorify, "_absorb_pending", "absorb")
wrap(CAT.Categorify, "fit_end", "cat_fit_end")
wrap(K, "fill_normalize_many", "fill_norm_launch")
wrap(nvt.Workflow, "fit", "fit")
wrap(nvt.Workflow, "transform", "transform")

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
frame = bench.synth_criteo(int(os.environ.get("ROWS", 45_000_000)), dev)
cats = [c for c in frame.columns if c.startswith("C")]
conts = [c for c in frame.columns if c.startswith("I")]
with tempfile.TemporaryDirectory() as tmp:
    wf = bench.build_workflow(cats, conts, os.path.join(tmp, "wf"))
    ds = nvt.Dataset(frame)
    for _ in range(4):
        wf.fit(ds)
        out = wf.transform(frame)
    torch.cuda.synchronize()
    del marks[:]
    for _ in range(20):
        wf.fit(ds)
        out = wf.transform(frame)
    torch.cuda.synchronize()
ev = sorted(marks, key=lambda m: m[1])
acc = {}
last_rs = None
for tag, t0, t1 in ev:
    if tag == "read_states":
        last_rs = t1
    acc.setdefault(tag, []).append(t1 - t0)
    if tag == "fill_norm_launch" and last_rs is not None:
        acc.setdefault("read_states_end -> fill_norm launched", []).append(t1 - last_rs)
        last_rs = None
# (absorb / cat_fit_end contain the read-back wait: subtract it)
for k, v in acc.items():
    per_step = sum(v) / 20.0
    v = sorted(v)
    print("%-40s per step %8.1f us   " % (k, 1e6 * per_step), end="")
    print("%-40s n %3d  median %8.1f us  min %8.1f  max %8.1f" % (k, len(v), 1e6 * v[len(v) // 2], 1e6 * v[0], 1e6 * v[-1]))
