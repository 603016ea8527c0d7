"""Micro-probe: nvt_dense_count paths by cardinality (HIP events per call)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvtabular_amd import kernels as K
if os.environ.get("NVT_PATH_TINY_MAX"):
    K.PATH_TINY_MAX = int(os.environ["NVT_PATH_TINY_MAX"])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 45_000_000
dev = torch.device("cuda", 0)
CASES = [(3, 1.1), (36, 1.1), (155, 1.1), (976, 1.1), (3000, 1.15), (7420, 1.1), (12972, 1.1), (20263, 1.1), (39043, 1.1), (403346, 1.2),
                (2953546, 1.15), (39884406, 1.05)]
if os.environ.get("PROBE_CARDS"):
    want = {int(x) for x in os.environ["PROBE_CARDS"].split(",")}
    CASES = [c for c in CASES if c[0] in want]
if os.environ.get("PROBE_S"):
    CASES = [(c, float(os.environ["PROBE_S"])) for c, _ in CASES]
for card, s in CASES:
    g = torch.Generator(device=dev).manual_seed(card)
    u = torch.rand(n, device=dev, dtype=torch.float64, generator=g)
    c = float(card)
    x = ((c ** (1.0 - s) - 1.0) * u + 1.0) ** (1.0 / (1.0 - s))
    keys = ((x.floor().clamp_(1, c).to(torch.int64) * 2654435761) % (2**31)).to(torch.int32)
    del u, x
    hint = 0
    for it in range(2):
        k, cnt, nulls, info = K.dense_count(keys, None, None, hint=hint)
        hint = info["distinct"]
    torch.cuda.synchronize()
    K.profile_begin()
    k, cnt, nulls, info = K.dense_count(keys, None, None, hint=hint)
    prof = K.profile_end()
    ms = sum(v[0] for v in prof.values())
    print(f"card={card:9d} distinct={hint:9d} path={info['path']} dense_count={ms*1e3:8.1f} us "
          f"{n*4/ms/1e6:7.1f} GB/s", flush=True)
