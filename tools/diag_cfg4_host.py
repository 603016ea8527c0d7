"""Host profile of the single-partition cfg4 step (TargetEncoding + JoinGroupby, 20 M rows)."""
import cProfile, pstats, io, os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvtabular_amd as nvt
from nvtabular_amd import ops
from nvtabular_amd.device import DeviceColumn, DeviceFrame

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
rows, card = 20_000_000, 5_000_000
g = torch.Generator(device=dev).manual_seed(7)
raw = (torch.rand(rows, device=dev, generator=g, dtype=torch.float64) ** 3 * card).to(torch.int64)
key = ((raw * 2654435761) % (2**31)).to(torch.int32)
y = torch.rand(rows, device=dev, generator=g, dtype=torch.float32)
frame = DeviceFrame({"k": DeviceColumn(key), "y": DeviceColumn(y)})
with tempfile.TemporaryDirectory() as tmp:
    te = ["k"] >> ops.TargetEncoding("y", kfold=5, fold_seed=42, p_smooth=20.0, defer_artifacts=True, out_path=os.path.join(tmp, "te"))
    jg = ["k"] >> ops.JoinGroupby(cont_cols=["y"], stats=["count", "sum", "mean", "std"], defer_artifacts=True, out_path=os.path.join(tmp, "jg"))
    wf = nvt.Workflow(te + jg)
    ds = nvt.Dataset(frame)

    def step():
        wf.fit(ds)
        return wf.transform(frame)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print("ms/step", 1e2 * (time.perf_counter() - t0))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10):
        step()
    torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:5000])
