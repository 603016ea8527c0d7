"""Host-side step-time outliers: run N bench steps, print the largest step periods and where they
were spent (fit / transform enqueue), plus allocator statistics before / after."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, nvtabular_amd as nvt
n = 45_000_000
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
frame = bench.synth_criteo(n, dev)
cats = [c for c in frame.columns if c.startswith("C")]
conts = [c for c in frame.columns if c.startswith("I")]
wf = bench.build_workflow(cats, conts, tempfile.mkdtemp())
ds = nvt.Dataset(frame)
import gc
for _ in range(3):
    wf.fit(ds); out = wf.transform(frame)
torch.cuda.synchronize(); gc.collect(); gc.disable()
st0 = torch.cuda.memory_stats()
marks = []
t_begin = time.perf_counter()
for i in range(steps):
    a = time.perf_counter(); wf.fit(ds); b = time.perf_counter(); out = wf.transform(frame); c = time.perf_counter()
    marks.append((a, b, c))
torch.cuda.synchronize()
total = time.perf_counter() - t_begin
st1 = torch.cuda.memory_stats()
per = [(marks[i + 1][0] - marks[i][0]) * 1e3 for i in range(steps - 1)]
print("steps", steps, "avg ms", 1e3 * total / steps, "median", sorted(per)[len(per) // 2])
top = sorted(range(len(per)), key=lambda i: -per[i])[:8]
for i in top:
    a, b, c = marks[i]
    print(f"step {i}: period {per[i]:.1f} ms  fit {1e3*(b-a):.1f}  transform {1e3*(c-b):.1f}")
for k in ("num_alloc_retries", "num_ooms", "num_device_alloc", "num_device_free", "reserved_bytes.all.peak", "allocated_bytes.all.peak"):
    print(k, st0.get(k), "->", st1.get(k))
