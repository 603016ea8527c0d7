"""One dense-id bench column through the range path with quantile splitters: the piecewise map
the sample kernel decided on, and -- recomputed on the host from it -- rows and distinct keys per
bucket against the capacities of the partition regions and the bucket tables."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from nvtabular_amd import _lib  # noqa: E402
from nvtabular_amd import kernels as K  # noqa: E402

j = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 45_000_000
hint = int(sys.argv[3]) if len(sys.argv) > 3 else 6_200_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(20260923 + j)
card = float(min(bench.CRITEO_CARDS[j], n))
s = [1.05, 1.1, 1.15, 1.2][j % 4]
u = torch.rand(n, device=dev, dtype=torch.float64, generator=g)
ids = (((card ** (1.0 - s) - 1.0) * u + 1.0) ** (1.0 / (1.0 - s))).floor().clamp_(1, card).to(torch.int32)
del u
ref = K.DenseCountJob(ids, None, None, hint=hint)
ref.path = K.PATH_SORT
rk, rc, _, rinfo = K.dense_count_many([ref])[0]
sp = K.range_splitters(rk, rc)
job = K.DenseCountJob(ids, None, None, hint=rinfo["distinct"], pieces=sp)
job.path = K.PATH_RANGE
batch = K.CountBatch([job])
st = K.read_back(batch.states)[0].tolist()
aux = job.hot_image.cpu().numpy().view(np.uint32)
print("overflow word", st[_lib.ST_OVERFLOW], "occupied", st[_lib.ST_OCCUPIED], "bits", job.range_bits())
LO, PW = 8192, 13600
S = int(aux[LO + 7])
print("piece_slots", S, "lo", aux[LO], "span", aux[LO + 1])
if S:
    P = 64
    spl = aux[PW:PW + P + 1].astype(np.int64)
    mul = aux[PW + P + 1:PW + 2 * P + 1].astype(np.uint64)
    fl = aux[PW + 2 * P + 1:PW + 2 * P + 3]
    hi = np.array([(int(fl[p >> 5]) >> (p & 31)) & 1 for p in range(P)], dtype=bool)
    print("splitters (unbiased)", (spl - 2**31)[:10], "...", (spl - 2**31)[-4:])
    print("widths", np.diff(spl)[:10], "...", np.diff(spl)[-4:])
    hk = ids.cpu().numpy()
    uk = (hk.astype(np.int64) + 2**31)
    img = aux[:8192].view(np.int32)
    hot = np.isin(hk, img[img != np.iinfo(np.int32).min])
    print("hot rows", hot.mean())
    p = np.clip(np.searchsorted(spl, uk, side="right") - 1, 0, P - 1)
    d = np.clip(uk - spl[p], 0, spl[p + 1] - spl[p] - 1).astype(np.uint64)
    f = np.where(hi[p], (d * mul[p]) >> np.uint64(32), (d * mul[p]) >> np.uint64(16)).astype(np.int64)
    f = np.minimum(f, S - 1) + p * S
    b = f >> 14
    NB = 1 << job.range_bits()
    cold_rows = np.bincount(b[~hot], minlength=NB)
    keys_u, first = np.unique(hk, return_index=True)
    distinct = np.bincount(b[first], minlength=NB)
    rows_per_wg = (n + 255) // 256
    cap = 2 * (rows_per_wg >> job.range_bits()) + 64
    print("cold rows per bucket: mean %.0f max %d (region cap per workgroup %d -> per bucket %d)" % (
        cold_rows.mean(), cold_rows.max(), cap, cap * 256))
    print("distinct per bucket: mean %.0f max %d (table takes 12288)" % (distinct.mean(), distinct.max()))
    worst = np.argsort(-distinct)[:8]
    print("worst buckets by distinct", [(int(x), int(distinct[x]), int(cold_rows[x])) for x in worst])
    worst = np.argsort(-cold_rows)[:8]
    print("worst buckets by cold rows", [(int(x), int(distinct[x]), int(cold_rows[x])) for x in worst])
    # slab-level: rows of ONE workgroup slab per bucket
    slab = hk.size // 256
    for w in (0, 100, 255):
        sl = slice(w * slab, (w + 1) * slab)
        cr = np.bincount(b[sl][~hot[sl]], minlength=NB)
        print("slab", w, "max cold rows in a bucket", cr.max(), "cap", cap)
