"""Steady-state launches of the range path on single bench columns (for rocprofv3 traces)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nvtabular_amd import kernels as K
dev = torch.device("cuda", 0)
rows = int(os.environ.get("ROWS", 45_000_000))
frame = bench.synth_criteo(rows, dev, n_cat=26, n_cont=0)
cols = [("C1", 6222134), ("C11", 1565431), ("C12", 368325), ("C2", 39042), ("C15", 11937)]
for name, hint in cols:
    c = frame[name]
    for it in range(3):
        job = K.DenseCountJob(c.data, c.valid, None, hint=hint)
        job.path = K.PATH_RANGE
        batch = K.CountBatch([job])
        torch.cuda.synchronize()
        st = batch.states.cpu().tolist()[0]
        k, cnt, nn, info = batch.results()[0]
    torch.cuda.synchronize()
    print(name, info["path"], info["distinct"], "cycles gather/hot/pass1/lookback/prefix/emit:", st[10:16], flush=True)
