#!/usr/bin/env python3
"""Launch timeline from a rocprofv3 kernel trace: the launches from the LAST occurrence of a marker kernel on, each with
start, the idle gap in front of it (no kernel of any stream running) and its duration; totals of busy and idle time.
usage: timeline.py <trace dir> <marker kernel substring> [--all]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
rows = rows[starts[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
busy = gaps = 0
prev_end = t0
big = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - prev_end)
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:46]
    if "--all" in sys.argv:
        print(f"{(s - t0) / 1e3:9.1f} us  +{gap / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {name}")
    busy += max(0, e - max(s, prev_end))
    gaps += gap
    big += gap > 4000
    prev_end = max(prev_end, e)
print(f"launches {len(rows)}  span {(prev_end - t0) / 1e3:.1f} us  busy {busy / 1e3:.1f} us  idle {gaps / 1e3:.1f} us  gaps > 4 us: {big}")
