"""How often the range path overflows on one bench column over many seeds, and why
(NVT_OVF_* bits).  usage: python tools/range_fail_probe.py [column index j] [seeds] [rows]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from nvtabular_amd import kernels as K  # noqa: E402

j = int(sys.argv[1]) if len(sys.argv) > 1 else 21
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 24
n = int(sys.argv[3]) if len(sys.argv) > 3 else 45_000_000
dev = torch.device("cuda", 0)
exps = [1.05, 1.1, 1.15, 1.2]
hint = None
for sd in range(seeds):
    g = torch.Generator(device=dev).manual_seed(31337 + 1000 * sd + j)
    card = float(min(bench.CRITEO_CARDS[j], n))
    s = exps[j % 4]
    u = torch.rand(n, device=dev, dtype=torch.float64, generator=g)
    x = (((card ** (1.0 - s) - 1.0) * u + 1.0) ** (1.0 / (1.0 - s))).floor().clamp_(1, card).to(torch.int64)
    ids = ((x * 2654435761 + 97 * j) % (2**31)).to(torch.int32)
    del u, x
    job = K.DenseCountJob(ids, None, None, hint=hint or 4_500_000)
    job.path = K.PATH_RANGE
    k, c, nn, info = K.dense_count_many([job])[0]
    hint = info["distinct"]
    print(sd, "path", info["path"], "distinct", info["distinct"], "failed", info.get("range_failed"),
          "bits", info.get("range_fail_bits"), "max", info["max_count"], flush=True)
