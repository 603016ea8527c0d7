"""Throughput of nvt_merge_sorted_many on bench-sized lists: 13 pairs of key-sorted (key, count)
lists of `n` entries each, `overlap` of the keys shared.  Prints ms and GB/s (12 B read per input
entry + 12 B written per output entry)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from nvtabular_amd import kernels as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 13
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
pairs = []
for j in range(npairs):
    def lst():
        k = torch.unique(torch.randint(-2**31, 2**31 - 1, (int(n * 1.3),), device=dev, generator=g,
                                       dtype=torch.int64) // 3 * 3)[:n].to(torch.int32)
        return k.contiguous(), torch.randint(1, 100, (k.numel(),), device=dev, generator=g, dtype=torch.int64)
    pairs.append((lst(), lst()))
res = K.merge_sorted_pairs(pairs)
tot_in = sum(a[0].numel() + b[0].numel() for a, b in pairs)
tot_out = sum(r[0].numel() for r in res)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
e0.record()
for _ in range(reps):
    K.merge_sorted_pairs(pairs)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("lib", os.environ.get("NVT_HIP_LIB", "default"), "in", tot_in, "out", tot_out,
      "ms %.3f" % ms, "GB/s %.0f" % ((tot_in + tot_out) * 12 / ms / 1e6))
