"""Phase timers of rp_count_kernel (library variant built with -DNVT_RANGE_TIMING: thread 0 of every
workgroup stamps clock64() between the phases and the deltas are summed into state[10..15]):
average cycles per workgroup and phase for the chosen bench columns.
NVT_HIP_LIB=.../libnvt_hip_timing.so python tools/rp_phase_probe.py [rows] C1 C2 ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from nvtabular_amd import kernels as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 45_000_000
names = [a for a in sys.argv[1:] if a.startswith("C")] or ["C1", "C2", "C20"]
dev = torch.device("cuda", 0)
frame = bench.synth_criteo(n, dev, n_cont=0)
phases = ["gather+insert", "hot keys", "E + look-back", "occ prefix", "emission loop (wave 0)", "max + last barrier"]
for name in names:
    col = frame[name]
    hint = 0
    for r in range(3):
        job = K.DenseCountJob(col.data, col.valid, None, hint=hint)
        res = K.dense_count_many([job])[0]
        hint = int(res[0].numel())
        torch.cuda.synchronize()
    st = job.state.cpu().tolist()
    nb = 1 << job.range_bits() if job.path == K.PATH_RANGE else 0
    if not nb:
        print(name, "not on the range path")
        continue
    cyc = [st[10 + q] / nb for q in range(6)]
    tot = sum(cyc)
    print(name, "distinct", hint, "buckets", nb, "cycles per workgroup:", " | ".join(
        f"{p} {c:.0f} ({100 * c / tot:.0f}%)" for p, c in zip(phases, cyc)), "total", round(tot))
