#!/bin/bash
# host API calls against kernel starts for the last GrayReorder call of reorder_cli (banded +-64): is the device stage's
# time in front of the first kernel (submission -> start) or behind the last one (completion -> host)?
export TMPDIR=/tmp
N=$((1 << 22))
python3 -c "
import sys; sys.path.insert(0, '.')
from sparsebase_amd import synth
rp, col = (t.cpu().numpy() for t in synth.banded_symmetric_torch($N, 64, per_row=12, seed=2))
rp.tofile('/tmp/g_rp.bin'); col.tofile('/tmp/g_col.bin')"
rm -rf /tmp/gray_kt2
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d /tmp/gray_kt2 -o kt -- sparsebase_amd/host/bin/reorder_cli gray /tmp/g_rp.bin /tmp/g_col.bin /tmp/g_out.bin $N $N 32 10 4 --device --time
ls /tmp/gray_kt2/*/ | head
python3 - <<'PY'
import csv, glob
kt = glob.glob("/tmp/gray_kt2/**/*_kernel_trace.csv", recursive=True)[0]
ht = glob.glob("/tmp/gray_kt2/**/*_hip_api_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(kt)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "GPU  " + r["Kernel_Name"][:50]))
for r in csv.DictReader(open(ht)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "host " + r["Function"]))
ev.sort()
# the last k_gray_rows_short kernel and everything within 60 ms before / 10 ms after
last = [e for e in ev if "k_gray_rows_short" in e[2]][-1][0]
t0 = None
for s, e, nm in ev:
    if last - 45e6 < s < last + 12e6:
        if t0 is None: t0 = s
        print(f"{(s - t0) / 1e3:10.1f} us  {(e - s) / 1e3:9.1f} us  {nm}")
PY
