"""encode_pipe_kernel, library variant built with -DNVT_ENC_PIPE_TIMING: per workgroup, cycles between the
first wave's end and the last wave's, and from the start of the loop to the last wave's end."""
import ctypes as C
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import nvtabular_amd as nvt  # noqa: E402
from nvtabular_amd import kernels as K  # noqa: E402
from nvtabular_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
frame = bench.synth_criteo(45_000_000, dev, n_cont=0)
lib = K._lib.load()
cols = [a for a in sys.argv[1:] if a.startswith("C")] or ["C1", "C2", "C11", "C23"]
with tempfile.TemporaryDirectory() as tmp:
    for c in cols:
        wf = nvt.Workflow([c] >> ops.Categorify(out_path=os.path.join(tmp, c), defer_artifacts=True))
        sub = frame[[c]]
        wf.fit(nvt.Dataset(sub))
        wf.transform(sub)
        torch.cuda.synchronize()
        v = (C.c_uint64 * 2)()
        K.check(lib.nvt_encode_stats(v, 1, K.stream_ptr()), "nvt_encode_stats")
        wf.transform(sub)
        K.check(lib.nvt_encode_stats(v, 1, K.stream_ptr()), "nvt_encode_stats")
        print(c, "per workgroup (256): first-to-last wave end %.0f cycles, loop %.0f cycles" % (v[0] / 256, v[1] / 256))
