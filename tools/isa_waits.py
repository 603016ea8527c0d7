#!/usr/bin/env python3
"""isa_waits.py <file.hip> <kernel substring> [-v] [extra hipcc flags...]: global loads / stores and the
`s_waitcnt vmcnt(N)` instructions of one kernel's ISA (device code only; runs on the CPU box).

A software-pipelined loop shows vmcnt(N > 0) in front of the first use of the PREVIOUS round's
loads; vmcnt(0) right behind every load means nothing is in flight across the round."""
import re
import subprocess
import sys

src, flt = sys.argv[1], sys.argv[2]
rest = sys.argv[3:]
verbose = "-v" in rest
extra = [a for a in rest if a != "-v"]
out = "/tmp/isa_waits.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                "-Wno-unused-function", "--cuda-device-only", "-S", src, "-o", out] + extra,
               check=True, capture_output=True)
text = open(out).read().split("\n")
start = None
for i, l in enumerate(text):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        if start is not None:
            break
        if flt in name:
            start = i
            kname = name
    if start is not None and l.startswith(".Lfunc_end"):
        break
end = i
body = text[start:end]
print(kname[:140])
ninst = sum(1 for l in body if l.startswith("\t") and not l.startswith("\t.") and not l.strip().startswith(";"))
print("instructions", ninst)
kinds = {}
for l in body:
    t = l.strip().split()
    if not t:
        continue
    op = t[0]
    if op.startswith(("global_load", "global_store", "buffer_load", "buffer_store", "flat_load", "flat_store",
                      "global_atomic", "scratch_")):
        kinds[op] = kinds.get(op, 0) + 1
print(kinds)
waits = [(j, l.strip()) for j, l in enumerate(body) if "s_waitcnt" in l and "vmcnt" in l]
hist = {}
for _, w in waits:
    m = re.search(r"vmcnt\((\d+)\)", w)
    hist[int(m.group(1))] = hist.get(int(m.group(1)), 0) + 1
print("vmcnt waits:", dict(sorted(hist.items())))
print("barriers", sum(1 for l in body if "s_barrier" in l), " s_cbranch", sum(1 for l in body if "s_cbranch" in l))
if verbose:
    for j, l in enumerate(body):
        s = l.strip()
        if s.startswith(("global_load", "global_store", "flat_load", "buffer_load")) or ("s_waitcnt" in s and "vmcnt" in s) \
                or s.startswith(".LBB") or "s_barrier" in s or "s_cbranch" in s:
            print(f"{j:6d} {s[:110]}")
