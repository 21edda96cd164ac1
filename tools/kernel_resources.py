#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of libw2a.so from hipcc's -Rpass-analysis=kernel-resource-usage.
usage: python tools/kernel_resources.py [extra hipcc flags ...]   (cross-compiles; no GPU needed)"""
import re
import subprocess
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from weather2alert_amd import build  # noqa: E402

cmd = [build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-I{build.INC}",
       build.SRC, "-o", "/tmp/_w2a_res.so", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[a-z/A-Z]+\])?: (\d+) \[-Rpass", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
demangle = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':70s} VGPR AGPR SGPR spill LDS   waves/SIMD")
for name, k in zip(demangle, rows.values()):
    if "rocprim" in name:
        continue
    print(f"{name[:70]:70s} {k.get('VGPRs', -1):4d} {k.get('AGPRs', -1):4d} {k.get('TotalSGPRs', -1):4d} "
          f"{k.get('VGPRs Spill', -1) + k.get('ScratchSize', 0):5d} {k.get('LDS Size', -1):5d} {k.get('Occupancy', -1):3d}")
