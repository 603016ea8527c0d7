"""Every categorical column of the bench frame counted ALONE (one column per CountBatch, nothing
else on the GPU), under `rocprofv3 --kernel-trace`: a marker kernel (nvt_fold_mt19937 of j + 2
rows) separates the columns in the trace.  tools/count_cols.sh runs it and prints, per column,
the kernels of one counting pass with their durations.
python tools/count_cols_probe.py [rows] [repeats]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from nvtabular_amd import _lib  # noqa: E402
from nvtabular_amd import kernels as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 45_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
lib = _lib.load()
frame = bench.synth_criteo(n, dev, n_cont=0)
mark = torch.empty(1024, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for j, (name, col) in enumerate(frame.items()):
    hint = None
    for r in range(reps + 1):
        if r == 1:   # the first pass sized the path (hint) and sampled the hot image
            torch.cuda.synchronize()
            rc = lib.nvt_fold_mt19937(1, 2, j + 2, mark.data_ptr(), s)
            assert rc == 0, rc
        job = K.DenseCountJob(col.data, col.valid, None, hint=hint if hint is not None else 0)
        res = K.dense_count_many([job])[0]
        hint = int(res[3]["distinct"]) if isinstance(res[3], dict) and "distinct" in res[3] else int(res[0].numel())
        torch.cuda.synchronize()
    lib.nvt_fold_mt19937(1, 2, 100 + j, mark.data_ptr(), s)   # end of this column's timed passes
    torch.cuda.synchronize()
    print(name, "distinct", hint, "path", job.path, "bits", job.range_bits() if job.path == K.PATH_RANGE else "-", flush=True)
