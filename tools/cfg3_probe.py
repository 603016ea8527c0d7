"""bench.py's cfg3 entry alone (4 high-cardinality columns), alone-mode per-kernel times."""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
res = bench.extra_cfg3(torch.device("cuda", 0), tempfile.mkdtemp(), int(os.environ.get("ROWS", 45_000_000)))
print(json.dumps({k: v for k, v in res.items() if k != "workload"}))
