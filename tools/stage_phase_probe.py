"""Phase timers of lds_stage_kernel (library variant built with -DNVT_STAGE_TIMING): average cycles per
workgroup for [table clear, main loop (wave 0), wait for the other waves, flush]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from nvtabular_amd import kernels as K  # noqa: E402

names = [a for a in sys.argv[1:] if a.startswith("C")] or ["C4", "C6", "C9", "C14"]
dev = torch.device("cuda", 0)
frame = bench.synth_criteo(45_000_000, dev, n_cont=0)
phases = ["clear", "main loop (wave 0)", "wait for the other waves", "flush"]
for name in names:
    col = frame[name]
    hint = 0
    for r in range(3):
        job = K.DenseCountJob(col.data, col.valid, None, hint=hint)
        res = K.dense_count_many([job])[0]
        hint = int(res[0].numel())
        torch.cuda.synchronize()
    st = job.state.cpu().tolist()
    cyc = [st[10 + q] / 256 for q in range(4)]
    tot = sum(cyc)
    print(name, "distinct", hint, "path", job.path, "cycles per workgroup:", " | ".join(
        f"{p} {c:.0f} ({100 * c / max(tot, 1):.0f}%)" for p, c in zip(phases, cyc)), "total", round(tot))
