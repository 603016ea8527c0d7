import sys, os, time, tempfile, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
import numpy as np, pandas as pd, torch
import bench, nvtabular_amd as nvt, oracle as O
from nvtabular_amd import ops, kernels as K
from nvtabular_amd.device import DeviceColumn, DeviceFrame
dev = torch.device("cuda", 0)
rows, card, p = 20_000_000, 5_000_000, 20.0
g = torch.Generator(device=dev).manual_seed(7)
raw = (torch.rand(rows, device=dev, generator=g, dtype=torch.float64) ** 3 * card).to(torch.int64)
key = ((raw * 2654435761) % (2**31)).to(torch.int32)
y = torch.rand(rows, device=dev, generator=g, dtype=torch.float32)
frame = DeviceFrame({"k": DeviceColumn(key), "y": DeviceColumn(y)})
stats = ["count", "sum", "mean", "std"]
tmp = tempfile.mkdtemp()
te = ["k"] >> ops.TargetEncoding("y", kfold=5, fold_seed=42, p_smooth=p, out_path=tmp + "/te", defer_artifacts=True)
jg = ["k"] >> ops.JoinGroupby(cont_cols=["y"], stats=stats, out_path=tmp + "/jg", defer_artifacts=True)
wf = nvt.Workflow(te + jg)
ds = nvt.Dataset(frame)
wf.fit(ds); out = wf.transform(frame); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); wf.fit(ds); torch.cuda.synchronize(); t1 = time.perf_counter(); out = wf.transform(frame); torch.cuda.synchronize(); t2 = time.perf_counter()
pr.disable()
print("fit ms", 1e3 * (t1 - t0), "transform ms", 1e3 * (t2 - t1))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:5000])
m = 500_000
sub = frame.slice_rows(0, m)
hdf = pd.DataFrame({"k": key[:m].cpu().numpy(), "y": y[:m].cpu().numpy()})
te_stats, te_means = O.target_encoding_fit([hdf.copy()], ["k"], ["y"], tmp + "/cte", kfold=5, fold_seed=42)
te_out = O.target_encoding_transform(hdf.copy(), ["k"], ["y"], te_stats, te_means, kfold=5, fold_seed=42, p_smooth=p)
jg_cats = O.join_groupby_fit([hdf.copy()], [["k"]], ["y"], stats, tmp + "/cjg")
jg_out = O.join_groupby_transform(hdf.copy(), [["k"]], jg_cats)
te2 = ["k"] >> ops.TargetEncoding("y", kfold=5, fold_seed=42, p_smooth=p, out_path=tmp + "/te2")
jg2 = ["k"] >> ops.JoinGroupby(cont_cols=["y"], stats=stats, out_path=tmp + "/jg2")
wf2 = nvt.Workflow(te2 + jg2); wf2.fit(nvt.Dataset(sub)); got = wf2.transform(sub)
print("gpu cols", got.columns, "oracle", list(te_out.columns), list(jg_out.columns))
for name, exp in [(c, te_out[c]) for c in te_out.columns if c.startswith("TE_")] + [(c, jg_out[c]) for c in jg_out.columns]:
    gv = got[name].data.cpu().numpy().astype("float64"); ev = exp.to_numpy().astype("float64")
    nan_mismatch = int((np.isnan(gv) != np.isnan(ev)).sum()); ok = ~np.isnan(ev) & ~np.isnan(gv)
    print(name, got[name].data.dtype, exp.dtype, "nan mismatch", nan_mismatch, "max rel", float(np.max(np.abs(gv[ok]-ev[ok])/np.maximum(np.abs(ev[ok]),1e-3))))
