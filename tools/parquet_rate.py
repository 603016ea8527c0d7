"""Parquet-inclusive rate: the Criteo-shaped frame written as uncompressed parquet (row groups
of 2**20 rows), then Dataset(path) -> fit + transform with pyarrow decoding on the host
cores, pinned staging and side-stream copies (nvtabular_amd/io.py)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import torch
import bench
import nvtabular_amd as nvt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rg_per_part = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
frame = bench.synth_criteo(n, dev)
cats = [c for c in frame.columns if c.startswith("C")]
conts = [c for c in frame.columns if c.startswith("I")]
arrays = {}
for name, col in frame.items():
    vals = col.data.cpu().numpy()
    if col.valid is not None:
        bits = np.unpackbits(col.valid.cpu().numpy(), bitorder="little")[:n]
        arrays[name] = pa.array(vals, mask=(bits == 0))
    else:
        arrays[name] = pa.array(vals)
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "day0.parquet")
t0 = time.perf_counter()
pq.write_table(pa.table(arrays), path, row_group_size=1 << 20, compression=None, use_dictionary=False,
               write_statistics=False)
print(f"wrote {os.path.getsize(path)/1e9:.2f} GB parquet in {time.perf_counter()-t0:.1f} s "
      f"({os.cpu_count()} host cores)", flush=True)
del frame, arrays
torch.cuda.empty_cache()
wf = bench.build_workflow(cats, conts, tempfile.mkdtemp())
ds = nvt.Dataset(path, engine="parquet", row_groups_per_part=rg_per_part)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wf.fit(ds)
    t1 = time.perf_counter()
    for part in wf.transform(ds).to_iter():
        del part
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iter {it}: fit {1e3*(t1-t0):.0f} ms, transform {1e3*(t2-t1):.0f} ms, "
          f"{n/(t2-t0)/1e6:.1f} M rows/s parquet -> HBM -> fit+transform", flush=True)

# output leg: transform + write the result as parquet (one file per partition)
out_dir = os.path.join(tmp, "out")
torch.cuda.synchronize(); t0 = time.perf_counter()
wf.transform(ds).to_parquet(out_dir)
dt = time.perf_counter() - t0
size = sum(os.path.getsize(os.path.join(out_dir, f)) for f in os.listdir(out_dir))
print(f"transform + to_parquet: {1e3*dt:.0f} ms, {n/dt/1e6:.1f} M rows/s, {size/1e9:.2f} GB written", flush=True)
from nvtabular_amd import io as _io
print("  phases", {k: round(v, 3) for k, v in _io.LAST_TIMING.items()}, flush=True)
for rep in range(2):
    import shutil
    shutil.rmtree(out_dir)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wf.transform(ds).to_parquet(out_dir)
    dt = time.perf_counter() - t0
    print(f"again: {1e3*dt:.0f} ms, {n/dt/1e6:.1f} M rows/s", {k: round(v, 3) for k, v in _io.LAST_TIMING.items()}, flush=True)
