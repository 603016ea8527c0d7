"""How does the partitioned counting pipeline scale when only the rows outside a hot key set
are valid?  (Feasibility probe for a hot-filter front pass: hot rows masked out through the
validity bitmap, paths 1 / 2 run on the rest.)"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from nvtabular_amd import kernels as K  # noqa: E402
from nvtabular_amd.device import pack_bitmap_device  # noqa: E402

dev = torch.device("cuda:0")
n = 45_000_000
frame = bench.synth_criteo(n, dev, n_cat=26, n_cont=0)


def timed(f, reps=5):
    f()
    torch.cuda.synchronize()
    K.profile_begin()
    for _ in range(reps):
        f()
    rep = K.profile_report()["kernels"]
    return {k: round(v[0] / reps * 1e3, 1) for k, v in rep.items()}


for name in ("C1", "C2", "C11", "C12", "C21", "C23"):
    col = frame[name]
    keys = col.data
    uk, cnt = torch.unique(keys, return_counts=True)
    order = torch.argsort(cnt, descending=True)
    hot = uk[order[:14336]]
    hot_mass = float(cnt[order[:14336]].sum()) / n
    lut_sorted, _ = torch.sort(hot)
    pos = torch.searchsorted(lut_sorted, keys).clamp_(max=lut_sorted.numel() - 1)
    is_hot = lut_sorted[pos] == keys
    cold_bitmap = pack_bitmap_device(~is_hot)
    d = int(uk.numel())
    path = K._path_for(d)
    a = timed(lambda: K.dense_count(keys, None, None, hint=d))
    b = timed(lambda: K.dense_count(keys, cold_bitmap, None, hint=d))
    print(name, "distinct", d, "path", path, "hot mass %.3f" % hot_mass, "full", a, "cold only", b, flush=True)
    del uk, cnt, order, pos, is_hot
