#!/bin/bash
# Kernel trace of the 64-bit calls (tools/int64_probe.py): which kernels ran, and that none of them narrows or widens.
# usage (gpurun, repo root): tools/int64_trace.sh  -> gpurun_out/int64_probe.log, gpurun_out/int64_kernels.txt
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
python3 tools/int64_probe.py > "$OUT/int64_probe.log" 2>&1
rm -rf /tmp/i64t
REPS=2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/i64t -o t -- python3 tools/int64_probe.py > /dev/null 2>&1
python3 - "$OUT/int64_kernels.txt" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/i64t/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
with open(sys.argv[1], "w") as o:
    o.write("narrow / widen launches: %d\n" % sum(int(r["Calls"]) for r in rows if "k_narrow" in r["Name"] or "k_widen" in r["Name"]))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        if "at::" in r["Name"] or "rocclr" in r["Name"]: continue
        o.write("%6s calls  avg %9.1f us  %s\n" % (r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
cat "$OUT/int64_probe.log"; head -40 "$OUT/int64_kernels.txt"
