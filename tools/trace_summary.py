#!/usr/bin/env python3
"""Print the launches of selected kernels from a rocprofv3 kernel trace, last step only."""
import csv, glob, sys
d, pats = sys.argv[1], sys.argv[2].split(",")
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(p in r["Kernel_Name"] for p in pats)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
t0 = int(rows[-n]["Start_Timestamp"]) if len(rows) >= n else int(rows[0]["Start_Timestamp"])
for r in rows[-n:]:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(r["Kernel_Name"].replace("(anonymous namespace)::", "")[:44].ljust(44), r["Grid_Size_X"].rjust(9),
          f"{dur:9.1f} us  @{(int(r['Start_Timestamp']) - t0) / 1e3:9.0f}")
