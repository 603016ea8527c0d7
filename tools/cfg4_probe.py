"""cfg4-shaped TargetEncoding + JoinGroupby fit + transform, a few steps (for rocprofv3)."""
import sys, tempfile, time
sys.path.insert(0, "/root/repo")
import torch
import nvtabular_amd as nvt
from nvtabular_amd import ops
from nvtabular_amd.device import DeviceColumn, DeviceFrame
dev = torch.device("cuda", 0)
rows, card, p = 20_000_000, 5_000_000, 20.0
g = torch.Generator(device=dev).manual_seed(7)
raw = (torch.rand(rows, device=dev, generator=g, dtype=torch.float64) ** 3 * card).to(torch.int64)
key = ((raw * 2654435761) % (2**31)).to(torch.int32)
y = torch.rand(rows, device=dev, generator=g, dtype=torch.float32)
frame = DeviceFrame({"k": DeviceColumn(key), "y": DeviceColumn(y)})
tmp = tempfile.mkdtemp()
te = ["k"] >> ops.TargetEncoding("y", kfold=5, fold_seed=42, p_smooth=p, out_path=tmp + "/te", defer_artifacts=True)
jg = ["k"] >> ops.JoinGroupby(cont_cols=["y"], stats=["count", "sum", "mean", "std"], out_path=tmp + "/jg", defer_artifacts=True)
wf = nvt.Workflow(te + jg)
ds = nvt.Dataset(frame)
for _ in range(3):
    wf.fit(ds); out = wf.transform(frame)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(6):
    wf.fit(ds); out = wf.transform(frame)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 6 * 1e3)
