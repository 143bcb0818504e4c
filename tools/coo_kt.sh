#!/bin/bash
# kernel trace of the C2B COO constructor sort (last call) -> timeline
export TMPDIR=/tmp
rm -rf /tmp/coo_kt
COO_PROBE_ONLY=c2b rocprofv3 --kernel-trace --output-format csv -d /tmp/coo_kt -o kt -- python3 tools/coo_sort_probe.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/coo_kt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last sort call: find last k_coo_is_sorted-like kernel start
names = [r['Kernel_Name'] for r in rows]
idx = max(i for i, n in enumerate(names) if 'sorted' in n.lower() or 'k_coo_check' in n.lower())
t0 = int(rows[idx]['Start_Timestamp'])
prev_end = t0
for r in rows[idx:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%8.1f us  gap %5.1f  %7.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:70]))
    prev_end = max(prev_end, e)
PY
