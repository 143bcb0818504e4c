#!/usr/bin/env python3
"""Register / LDS / spill table of the kernels of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kres.py sparsebase_amd/csrc/sbx_permute.hip [name-filter] [extra hipcc flags...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
       "-I", os.path.join(ROOT, "sparsebase_amd", "csrc"), "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[\w/]+\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print(f"{k[:70]:70s} VGPR {v.get('VGPRs', -1):4d} AGPR {v.get('AGPRs', -1):3d} SGPR {v.get('TotalSGPRs', -1):4d} "
              f"spill {v.get('VGPRs Spill', 0):3d} scratch {v.get('ScratchSize', 0):5d} occ {v.get('Occupancy', -1):2d} LDS {v.get('LDS Size', 0):6d}")
