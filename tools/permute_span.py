#!/usr/bin/env python3
"""One Permute2D of tools/permute_only.py in a rocprofv3 kernel trace (side streams on): span from the first to the last
kernel of the last call, time with at least one kernel running, idle gaps.  usage: permute_span.py <trace dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_rowwise_prep" in r["Kernel_Name"]]
rows = rows[starts[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
end = t0
busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    gap = s - end
    print(f"{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  {'gap %5.1f' % (gap / 1e3) if gap > 0 else '         '}  {name}")
    if s > end:
        busy += e - s
    elif e > end:
        busy += e - end
    end = max(end, e)
print(f"span {(end - t0) / 1e3:.1f} us, at least one kernel running {busy / 1e3:.1f} us, idle {(end - t0 - busy) / 1e3:.1f} us")
