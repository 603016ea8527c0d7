"""PCIe-inclusive rate: the same Criteo-shaped workload, but the
boundary is handed HOST Arrow tables (what the parquet reader produces), partition by
partition; Dataset.to_iter stages them through pinned memory on a side stream while the
previous partition is being processed.  fit + transform(+ drop outputs), wall clock."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyarrow as pa
import torch
import bench
import nvtabular_amd as nvt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 45_000_000
nparts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
frame = bench.synth_criteo(n, dev)
cats = [c for c in frame.columns if c.startswith("C")]
conts = [c for c in frame.columns if c.startswith("I")]
# device frame -> host Arrow tables (int32 + validity bitmaps), row ranges multiple of 8
step = (n // nparts) // 8 * 8
tables = []
for p in range(nparts):
    lo, hi = p * step, (n if p == nparts - 1 else (p + 1) * step)
    arrays = {}
    for name, col in frame.items():
        vals = col.data[lo:hi].cpu().numpy()
        if col.valid is not None:
            bits = np.unpackbits(col.valid[lo // 8:(hi + 7) // 8].cpu().numpy(), bitorder="little")[: hi - lo]
            arrays[name] = pa.array(vals, mask=(bits == 0))
        else:
            arrays[name] = pa.array(vals)
    tables.append(pa.table(arrays))
del frame
torch.cuda.empty_cache()
nbytes = sum(t.nbytes for t in tables)
wf = bench.build_workflow(cats, conts, tempfile.mkdtemp())
ds = nvt.Dataset(tables)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wf.fit(ds)
    t1 = time.perf_counter()
    for part in wf.transform(ds).to_iter():
        del part
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iter {it}: fit {1e3*(t1-t0):.0f} ms, transform {1e3*(t2-t1):.0f} ms, "
          f"{n/(t2-t0)/1e6:.1f} M rows/s, host input {nbytes/1e9:.2f} GB read twice -> "
          f"{2*nbytes/(t2-t0)/1e9:.1f} GB/s over PCIe", flush=True)
