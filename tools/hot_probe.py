"""Per-kernel cost of one partitioned column with and without the hot-key filter (run under
rocprofv3 --kernel-trace --stats)."""
import sys

import torch

sys.path.insert(0, "/root/repo")
import bench  # noqa: E402
from nvtabular_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
n = 45_000_000
frame = bench.synth_criteo(n, dev, n_cat=26, n_cont=0)
K.COUNT_STREAMS = 1
which = sys.argv[1]
hot = sys.argv[2] == "1"
keys = frame[which].data
d = int(torch.unique(keys).numel())
for _ in range(6):
    job = K.DenseCountJob(keys, None, None, hint=d)
    job.hot = hot
    K.dense_count_many([job])
torch.cuda.synchronize()
