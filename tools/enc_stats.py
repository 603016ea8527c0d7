"""Cache-mode encode: rows that miss the LDS head (and pay one random sector of the table in HBM)
per transform of the bench frame, per column.  Run once per head layout:
    NVT_ENC_STATS=1 NVT_ENC_HEAD16=0|1 python tools/enc_stats.py"""
import ctypes as C
import json
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import nvtabular_amd as nvt  # noqa: E402
from nvtabular_amd import kernels as K  # noqa: E402
from nvtabular_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
n = int(os.environ.get("ROWS", 45_000_000))
frame = bench.synth_criteo(n, dev)
cats = [c for c in frame.columns if c.startswith("C")]
lib = K._lib.load()
out = {"rows": n, "head16": os.environ.get("NVT_ENC_HEAD16", "1") != "0", "columns": {}}
tot_m = tot_r = 0
with tempfile.TemporaryDirectory() as tmp:
    for c in cats:
        wf = nvt.Workflow([c] >> ops.Categorify(out_path=os.path.join(tmp, c), defer_artifacts=True))
        sub = frame[[c]]
        wf.fit(nvt.Dataset(sub))
        wf.transform(sub)
        torch.cuda.synchronize()
        v = (C.c_uint64 * 2)()
        K.check(lib.nvt_encode_stats(v, 1, K.stream_ptr()), "nvt_encode_stats")
        wf.transform(sub)
        K.check(lib.nvt_encode_stats(v, 1, K.stream_ptr()), "nvt_encode_stats")
        if v[1]:
            out["columns"][c] = {"looked_up": int(v[1]), "missed_head": int(v[0]),
                                 "miss_frac": round(v[0] / v[1], 4)}
            tot_m += int(v[0])
            tot_r += int(v[1])
out["missed_head_rows_per_step"] = tot_m
out["looked_up_rows_per_step"] = tot_r
print(json.dumps(out))
