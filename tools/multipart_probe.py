"""bench.py's cfg3_multipartition entry alone (8 x 45 M rows by default), one JSON line.
usage: python tools/multipart_probe.py [rows] [nparts] [single_ms]"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 45_000_000
nparts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
single = float(sys.argv[3]) if len(sys.argv) > 3 else None
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
res = bench.extra_multipart(dev, tempfile.mkdtemp(prefix="nvt_mp_"), rows, nparts, single_ms=single)
print(json.dumps(res))
