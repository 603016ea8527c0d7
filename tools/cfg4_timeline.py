"""cfg4 single-partition step under `rocprofv3 --kernel-trace`: steps separated by 50 ms of sleep so
that tools/timeline_gaps.py can cut the trace into steps and list every launch with the idle gap in
front of it (where the host does not keep the device fed)."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvtabular_amd as nvt  # noqa: E402
from nvtabular_amd import ops  # noqa: E402
from nvtabular_amd.device import DeviceColumn, DeviceFrame  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
rows, card = 20_000_000, 5_000_000
g = torch.Generator(device=dev).manual_seed(7)
raw = (torch.rand(rows, device=dev, generator=g, dtype=torch.float64) ** 3 * card).to(torch.int64)
key = ((raw * 2654435761) % (2**31)).to(torch.int32)
y = torch.rand(rows, device=dev, generator=g, dtype=torch.float32)
frame = DeviceFrame({"k": DeviceColumn(key), "y": DeviceColumn(y)})
with tempfile.TemporaryDirectory() as tmp:
    te = ["k"] >> ops.TargetEncoding("y", kfold=5, fold_seed=42, p_smooth=20.0, defer_artifacts=True,
                                     out_path=os.path.join(tmp, "te"))
    jg = ["k"] >> ops.JoinGroupby(cont_cols=["y"], stats=["count", "sum", "mean", "std"], defer_artifacts=True,
                                  out_path=os.path.join(tmp, "jg"))
    wf = nvt.Workflow(te + jg)
    ds = nvt.Dataset(frame)
    for _ in range(4):
        wf.fit(ds)
        out = wf.transform(frame)
    torch.cuda.synchronize()
    for _ in range(3):
        time.sleep(0.05)
        t0 = time.perf_counter()
        wf.fit(ds)
        t1 = time.perf_counter()
        out = wf.transform(frame)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print("host: fit %.3f ms, transform enqueue %.3f ms, drain %.3f ms, total %.3f ms" % (
            1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0)))
