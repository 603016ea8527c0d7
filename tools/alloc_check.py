"""Does the steady-state step still reach the device allocator (hipMalloc / hipFree)?  Those
calls synchronise the device and take milliseconds; a step should be served entirely from
torch's caching allocator after the warm-up."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import nvtabular_amd as nvt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 45_000_000
dev = torch.device("cuda", 0)
frame = bench.synth_criteo(n, dev)
cats = [c for c in frame.columns if c.startswith("C")]
conts = [c for c in frame.columns if c.startswith("I")]
wf = bench.build_workflow(cats, conts, tempfile.mkdtemp())
ds = nvt.Dataset(frame)
out = None
for step in range(8):
    s0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wf.fit(ds); out = wf.transform(frame); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    s1 = torch.cuda.memory_stats()
    print(f"step {step}: {dt:6.2f} ms  segments +{s1['num_device_alloc'] - s0['num_device_alloc']} "
          f"-{s1['num_device_free'] - s0['num_device_free']}  reserved {s1['reserved_bytes.all.current']/2**30:.1f} GiB "
          f"retries {s1['num_alloc_retries']}", flush=True)
