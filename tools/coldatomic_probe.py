"""Rate of the global-table (device atomics) counting kernel on the COLD rows only (rows whose
key is outside the 14336 hottest keys) -- feasibility of "LDS hot filter + global cold table"."""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from nvtabular_amd import kernels as K  # noqa: E402
from nvtabular_amd.device import pack_bitmap_device  # noqa: E402

dev = torch.device("cuda:0")
n = 45_000_000
frame = bench.synth_criteo(n, dev, n_cat=26, n_cont=0)

for name in ("C3", "C5", "C2", "C12", "C23", "C11", "C1"):
    keys = frame[name].data
    uk, cnt = torch.unique(keys, return_counts=True)
    order = torch.argsort(cnt, descending=True)
    hot = uk[order[:14336]]
    lut_sorted, _ = torch.sort(hot)
    pos = torch.searchsorted(lut_sorted, keys).clamp_(max=lut_sorted.numel() - 1)
    is_hot = lut_sorted[pos] == keys
    cold_rows = int((~is_hot).sum())
    cold_bitmap = pack_bitmap_device(~is_hot)
    d = int(uk.numel())
    cold_d = max(d - 14336, 1)
    tab = K.CountTable(keys.dtype, max(4 * cold_d, 1 << 16))
    res = []
    for rep in range(4):
        tab.clear()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tab.update(keys, cold_bitmap)
        e1.record()
        torch.cuda.synchronize()
        res.append(round(e0.elapsed_time(e1) * 1e3, 1))
    print(name, "distinct", d, "cold rows %.2f M" % (cold_rows / 1e6), "cold distinct", cold_d,
          "capacity", tab.capacity, res[1:], flush=True)
    del uk, cnt, order, pos, is_hot, tab
