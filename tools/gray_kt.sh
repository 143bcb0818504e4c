#!/bin/bash
# Kernel trace of the LAST (timed) GrayReorder call of `reorder_cli gray ... --device --time` (exact mode, program directly
# behind `--`): where the device stage's milliseconds go — kernel durations against the gaps between them and against the
# time from the previous call's last kernel.  usage (gpurun, repo root): tools/gray_kt.sh -> gpurun_out/gray_kt_<case>.txt
export TMPDIR=/tmp
N=$((1 << 22))
for case in banded_w64 rmat22; do
  if [ $case = banded_w64 ]; then MAKE="synth.banded_symmetric_torch($N, 64, per_row=12, seed=2)"; else MAKE="synth.rmat_symmetric_torch(22, 13, seed=1)"; fi
  python3 -c "
import sys; sys.path.insert(0, '.')
from sparsebase_amd import synth
rp, col = (t.cpu().numpy() for t in $MAKE)
rp.tofile('/tmp/g_rp.bin'); col.tofile('/tmp/g_col.bin')"
  rm -rf /tmp/gray_kt_$case
  rocprofv3 --kernel-trace --output-format csv -d /tmp/gray_kt_$case -o kt -- sparsebase_amd/host/bin/reorder_cli gray /tmp/g_rp.bin /tmp/g_col.bin /tmp/g_out.bin $N $N 32 10 4 --device --time > gpurun_out/gray_kt_$case.stdout 2> gpurun_out/gray_kt_$case.stderr
  python3 - /tmp/gray_kt_$case > gpurun_out/gray_kt_$case.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# calls are separated by the host stage: gaps of more than 5 ms between kernels
calls, cur, prev_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None and s - prev_end > 5e6:
        calls.append(cur); cur = []
    cur.append((s, e, r["Kernel_Name"]))
    prev_end = e
calls.append(cur)
print("calls (kernel groups separated by > 5 ms):", len(calls))
for ci, c in enumerate(calls[-3:]):
    t0 = c[0][0]
    idle_before = None
    idx = len(calls) - 3 + ci
    if idx > 0 and calls[idx - 1]:
        idle_before = (t0 - calls[idx - 1][-1][1]) / 1e6
    print(f"--- call {idx}: {len(c)} kernels, span {(c[-1][1] - t0) / 1e3:.1f} us, GPU idle before it {idle_before} ms")
    pe = t0
    for s, e, nm in c:
        nm = nm.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        print(f"{(s - t0) / 1e3:9.1f} us  +{max(0, s - pe) / 1e3:7.1f} gap  {(e - s) / 1e3:8.1f} us  {nm}")
        pe = max(pe, e)
PY
  echo "== $case"; cat gpurun_out/gray_kt_$case.stdout; tail -30 gpurun_out/gray_kt_$case.txt
done
