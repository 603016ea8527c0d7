"""Sort-path reductions alone (K.sorted_groupby twice inside one pass: TargetEncoding's shape,
then JoinGroupby's) on skewed (bench cfg4) or uniform keys: which part of gb_segreduce_kernel's
time is the data's skew (same-address atomics of runs that span many rows of 64 words)?
SKEW=3 (bench) | 1 (uniform); ROWS, CARD."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvtabular_amd import kernels as K  # noqa: E402

dev = torch.device("cuda", 0)
rows, card = int(os.environ.get("ROWS", 20_000_000)), int(os.environ.get("CARD", 5_000_000))
skew = float(os.environ.get("SKEW", 3))
g = torch.Generator(device=dev).manual_seed(7)
raw = (torch.rand(rows, device=dev, generator=g, dtype=torch.float64) ** skew * card).to(torch.int64)
key = ((raw * 2654435761) % (2**31)).to(torch.int32)
y = torch.rand(rows, device=dev, generator=g, dtype=torch.float32)
fold = (torch.arange(rows, device=dev) % 5).to(torch.uint8)


def once():
    with K.pass_memo():
        a = K.sorted_groupby(key, fold, 5, [y], [None], te_records=True)
        b = K.sorted_groupby(key, None, 1, [y], [None], sumsq=True)
    return a, b


for _ in range(2):
    once()
torch.cuda.synchronize()
K.profile_begin()
t0 = time.perf_counter()
for _ in range(5):
    once()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
rep = K.profile_report()
print(f"skew {skew} rows {rows} card {card}: {1e3 * dt:.3f} ms per pass;",
      {k: round(v[0] / 5, 3) for k, v in rep["kernels"].items()})
