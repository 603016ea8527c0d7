"""How fast can this box take 14 GB of column buffers into files?  pwrite vs mmap, one file vs
several, thread counts.  usage: write_probe.py [GB] [dir]"""
import mmap
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
base = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp()
chunk = 32 << 20
nchunks = int(gb * (1 << 30) / chunk)
src = np.random.default_rng(0).integers(0, 255, chunk, dtype=np.uint8)
srcs = [src.copy() for _ in range(8)]


def run(name, nfiles, threads, mode):
    paths = [os.path.join(base, f"probe_{i}.bin") for i in range(nfiles)]
    fds = [os.open(p, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644) for p in paths]
    per = nchunks // nfiles
    maps = None
    if mode == "mmap":
        for fd in fds:
            os.ftruncate(fd, per * chunk)
        maps = [np.frombuffer(mmap.mmap(fd, per * chunk), dtype=np.uint8) for fd in fds]

    def job(i):
        f, c = i % nfiles, i // nfiles
        if mode == "pwrite":
            os.pwrite(fds[f], memoryview(srcs[i % 8]), c * chunk)
        else:
            np.copyto(maps[f][c * chunk:(c + 1) * chunk], srcs[i % 8])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(job, range(per * nfiles)))
    dt = time.perf_counter() - t0
    print(f"{name:28s} files {nfiles:2d} threads {threads:2d}: {per * nfiles * chunk / dt / 1e9:6.1f} GB/s", flush=True)
    maps = None
    for fd in fds:
        os.close(fd)
    for p in paths:
        os.remove(p)


for nfiles, threads in ((1, 1), (1, 16), (6, 6), (6, 16), (12, 24), (39, 39)):
    run("pwrite", nfiles, threads, "pwrite")
for nfiles, threads in ((1, 16), (6, 16)):
    run("mmap", nfiles, threads, "mmap")
