"""bench.py's end_to_end entry alone.  usage: e2e_probe.py [rows] [nparts]"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

a = [int(x) for x in sys.argv[1:]]
torch.cuda.set_device(0)
print(json.dumps(bench.extra_end_to_end(torch.device("cuda", 0), tempfile.mkdtemp(prefix="nvt_e2e_"),
                                        *(a or [45_000_000]))))
