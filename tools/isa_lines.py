#!/usr/bin/env python3
"""Static instruction histogram of one kernel by source line: tools/isa_lines.py <file.s (compiled with -g1 -S)> <mangled-substring> [bucket]
Prints, per source line (or bucket of lines), how many VALU / SALU / DS / VMEM instructions the kernel's code attributes to it."""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 1
files = {}
hist = collections.defaultdict(lambda: [0, 0, 0, 0, 0])
on = False
cur = (0, 0)
for line in open(path, errors="replace"):
    t = line.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', t)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    if re.match(r"^[A-Za-z_][\w$.]*:", t) and key in t.split(":")[0]:
        on = True
        continue
    if on and t.startswith(".amdhsa_kernel"):
        break
    if not on:
        continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        cur = (int(m.group(1)), int(m.group(2)) // bucket * bucket)
        continue
    if not t or t[0] in ".;" or re.match(r"^[A-Za-z_.][\w$.]*:", t):
        continue
    op = t.split()[0]
    k = 0 if op.startswith("v_") else 1 if op.startswith("s_") else 2 if op.startswith("ds_") else 3 if op.split("_")[0] in ("global", "buffer", "flat", "scratch") else 4
    hist[cur][k] += 1
tot = [0] * 5
for (f, l), v in sorted(hist.items()):
    print(f"{files.get(f, f)}:{l:5d}  VALU {v[0]:5d} SALU {v[1]:5d} DS {v[2]:4d} VMEM {v[3]:4d} other {v[4]:3d}")
    tot = [a + b for a, b in zip(tot, v)]
print("total", tot)
