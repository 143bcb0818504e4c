#!/usr/bin/env python3
"""Timeline of the LAST RCM call of tools/rcm_trace.py from a rocprofv3 kernel trace: every launch with its start,
duration and the idle gap in front of it; totals of busy and idle time.  usage: rcm_timeline.py <trace dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last call starts at the last k_cc_init (the first kernel of sbx_rcm_reorder)
starts = [i for i, r in enumerate(rows) if "k_deg_count" in r["Kernel_Name"] or "k_cc_init" in r["Kernel_Name"]]
first = starts[-2] if len(starts) >= 2 and "k_cc_init" in rows[starts[-1]]["Kernel_Name"] and "k_deg_count" in rows[starts[-2]]["Kernel_Name"] else starts[-1]
rows = rows[first:]
t0 = int(rows[0]["Start_Timestamp"])
busy = gaps = 0
prev_end = t0
big_gaps = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - prev_end)
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
    if "--all" in sys.argv:
        print(f"{(s - t0) / 1e3:9.1f} us  +{gap / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>2} s{r.get('Stream_Id', '?'):>2}  {name}")
    busy += max(0, e - max(s, prev_end))
    gaps += gap
    if gap > 4000:
        big_gaps.append((gap / 1e3, name))
    prev_end = max(prev_end, e)
print(f"launches {len(rows)}  span {(prev_end - t0) / 1e3:.1f} us  busy {busy / 1e3:.1f} us  idle {gaps / 1e3:.1f} us  gaps > 4 us: {len(big_gaps)} totalling {sum(g for g, _ in big_gaps):.1f} us")
