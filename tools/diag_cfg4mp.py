"""Host profile of one cfg4_multipartition step after the cfg4 single-partition entry ran in the
same process (the order of bench.py's extras)."""
import cProfile, pstats, io, os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import nvtabular_amd as nvt
from nvtabular_amd import ops

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
with tempfile.TemporaryDirectory() as tmp:
    if os.environ.get("FIRST", "1") == "1":
        bench.extra_cfg4(dev, tmp, 20_000_000, 200_000, steps=2)
    rows, nparts, card = 1 << 28, 4, 100_000_000

    def make(seed, n):
        g = torch.Generator(device=dev).manual_seed(seed)
        raw = (torch.rand(n, device=dev, generator=g, dtype=torch.float64) ** 3 * card).to(torch.int64)
        key = ((raw * 2654435761) % (2**31)).to(torch.int32)
        y = torch.rand(n, device=dev, generator=g, dtype=torch.float32)
        from nvtabular_amd.device import DeviceColumn, DeviceFrame
        return DeviceFrame({"k": DeviceColumn(key), "y": DeviceColumn(y)})

    frames = [make(700 + i, rows) for i in range(nparts)]
    te = ["k"] >> ops.TargetEncoding("y", kfold=5, fold_seed=42, p_smooth=20.0, defer_artifacts=True, out_path=os.path.join(tmp, "te"))
    jg = ["k"] >> ops.JoinGroupby(cont_cols=["y"], stats=["count", "sum", "mean", "std"], defer_artifacts=True, out_path=os.path.join(tmp, "jg"))
    wf = nvt.Workflow(te + jg)
    ds = nvt.Dataset(frames)

    def step():
        wf.fit(ds)
        for out in wf.transform(ds).to_iter():
            del out

    step(); torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); print("step ms", 1e3 * (time.perf_counter() - t0))
    pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
