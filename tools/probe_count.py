"""Micro-probe: Categorify.fit count kernel on columns of chosen cardinality.
Usage: python tools/probe_count.py [rows] ; prints per-cardinality kernel time (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvtabular_amd import kernels as K

n = int(sys.argv[1]) if len(sys.argv) > 1 else 45_000_000
dev = torch.device("cuda", 0)
for card, s in [(3, 1.1), (36, 1.1), (155, 1.2), (976, 1.1), (7120, 1.15), (39043, 1.1), (403346, 1.2),
                (2953546, 1.15), (39884406, 1.05)]:
    g = torch.Generator(device=dev).manual_seed(card)
    u = torch.rand(n, device=dev, dtype=torch.float64, generator=g)
    c = float(card)
    x = ((c ** (1.0 - s) - 1.0) * u + 1.0) ** (1.0 / (1.0 - s))
    keys = ((x.floor().clamp_(1, c).to(torch.int64) * 2654435761) % (2**31)).to(torch.int32)
    del u, x
    hint = 64
    for it in range(3):
        tab, st = K.count_into_new_table([keys], [None], hint)
        hint = st[2]
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tab = K.CountTable(torch.int32, 2 * hint)
    a.record(); tab.update(keys, None); b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    print(f"card={card:9d} uniques={hint:9d} count_kernel={ms*1e3:8.1f} us  {n*4/ms/1e6:7.1f} GB/s", flush=True)
