"""bench.py's cfg2_dense_ids entry alone.  usage: python tools/dense_ids_probe.py [rows]"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 45_000_000
torch.cuda.set_device(0)
res = bench.extra_dense_ids(torch.device("cuda", 0), tempfile.mkdtemp(prefix="nvt_di_"), rows,
                            single_ms=float(sys.argv[2]) if len(sys.argv) > 2 else None)
print(json.dumps(res))
