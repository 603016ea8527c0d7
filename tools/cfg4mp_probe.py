"""bench.py's cfg4_multipartition entry alone.  usage: cfg4mp_probe.py [rows_per_part] [nparts] [card]"""
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
res = bench.extra_cfg4_multipart(torch.device("cuda", 0), tempfile.mkdtemp(prefix="nvt_c4_"),
                                 *(a[:3] if a else []))
print(json.dumps(res))
