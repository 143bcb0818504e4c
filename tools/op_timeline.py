#!/usr/bin/env python3
"""Timeline of the LAST call of an operation from a rocprofv3 kernel trace: every launch from the last occurrence of the
kernel named by <first> on, with its start, duration and the idle gap in front of it.
usage: op_timeline.py <trace dir> <substring of the call's first kernel> [substring of the kernel that ends the call]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
rows = rows[starts[-1]:]
if len(sys.argv) > 3:
    ends = [i for i, r in enumerate(rows) if sys.argv[3] in r["Kernel_Name"]]
    if ends: rows = rows[:ends[-1] + 1]
t0 = int(rows[0]["Start_Timestamp"])
busy = gaps = 0
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - prev_end)
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {name}")
    busy += max(0, e - max(s, prev_end))
    gaps += gap
    prev_end = max(prev_end, e)
print(f"launches {len(rows)}  span {(prev_end - t0) / 1e3:.1f} us  busy {busy / 1e3:.1f} us  idle {gaps / 1e3:.1f} us")
