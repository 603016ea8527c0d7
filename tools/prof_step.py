"""cProfile one bench step (host-side overhead hunt)."""
import cProfile, pstats, sys, os, tempfile, io
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
for _ in range(2):
    wf.fit(ds); wf.transform(frame)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
wf.fit(ds); out = wf.transform(frame); torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
