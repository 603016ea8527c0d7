"""cfg4 (TargetEncoding + JoinGroupby, 20 M rows x 5 M keys) alone: ms per step + per-kernel times."""
import json
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
with tempfile.TemporaryDirectory() as tmp:
    res = bench.extra_cfg4(dev, tmp, int(os.environ.get("ROWS", 20_000_000)), 200_000, steps=5)
print(json.dumps(res, indent=1, default=str))
