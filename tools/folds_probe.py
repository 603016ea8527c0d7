"""Seeded fold generation: serial kernel vs parallel chunks (ms), 21 M and 2^28 + 1/16 values."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvtabular_amd.ops import target_encoding as T

dev = torch.device("cuda:0")
for n in (20_000_000, 1 << 28):
    for par in (True, False):
        if not par and n > 50_000_000:
            continue
        T.PARALLEL_FOLDS = par
        for rep in range(2):
            T._FOLD_CACHE.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            T._fold_column(n, 5, 42, dev)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0)
        print(f"n {n} parallel {par}: {ms:.2f} ms")
