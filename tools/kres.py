#!/usr/bin/env python3
"""kres.py <file.hip> [name filter] [extra hipcc flags...]: registers / scratch / LDS per kernel
(hipcc -Rpass-analysis=kernel-resource-usage, device code only; runs on the CPU box)."""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
       "--cuda-device-only", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\S+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    print(f"{name[:100]:100s} vgpr {v.get('VGPRs')} spill {v.get('VGPRs Spill')} sgpr {v.get('TotalSGPRs')} "
          f"sspill {v.get('SGPRs Spill')} scratch {v.get('ScratchSize')} lds {v.get('LDS Size')} occ {v.get('Occupancy')}")
