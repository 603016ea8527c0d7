"""Time the FIRST fit+transform of a fresh workflow (no learned cardinality hints) against
the steady state bench.py reports."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import nvtabular_amd as nvt
from nvtabular_amd import kernels as K

n = int(sys.argv[1]) if len(sys.argv) > 1 else 45_000_000
dev = torch.device("cuda", 0)
frame = bench.synth_criteo(n, dev)
cats = [c for c in frame.columns if c.startswith("C")]
conts = [c for c in frame.columns if c.startswith("I")]
# warm the allocator / library with a different workflow object on a small slice
small = frame.slice_rows(0, 1_000_000)
wf0 = bench.build_workflow(cats, conts, tempfile.mkdtemp())
wf0.fit(nvt.Dataset(small)); wf0.transform(small); torch.cuda.synchronize()
for trial in range(2):
    wf = bench.build_workflow(cats, conts, tempfile.mkdtemp())
    ds = nvt.Dataset(frame)
    for step in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K.profile_begin()
        wf.fit(ds); out = wf.transform(frame); torch.cuda.synchronize()
        prof = K.profile_end()
        dt = (time.perf_counter() - t0) * 1e3
        top = sorted(((v[0], k) for k, v in prof.items()), reverse=True)[:4]
        print(f"trial {trial} step {step}: {dt:7.2f} ms  " + ", ".join(f"{k}={t:.1f}" for t, k in top), flush=True)
        del out
