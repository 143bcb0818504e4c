#!/usr/bin/env python3
"""Runs reorder_cli many times on a few fixtures and reports crashes / hangs / differing outputs
(hunting process-level flakiness: each run is a fresh process with its own HIP context)."""
import hashlib, json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cli = os.path.join(ROOT, "sparsebase_amd", "host", "bin", "reorder_cli")
z = np.load(os.path.join(ROOT, "tests", "golden", "small_cases.npz"))
meta = json.load(open(os.path.join(ROOT, "tests", "golden", "small_cases.json")))
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tmp = tempfile.mkdtemp()
jobs = []
for name in ("rmat12", "sym_d", "grid_shuffled"):
    rp, col = z[f"{name}/row_ptr"], z[f"{name}/col"]
    n = len(rp) - 1
    a, b = os.path.join(tmp, name + ".rp"), os.path.join(tmp, name + ".col")
    rp.astype(np.int32).tofile(a); col.astype(np.int32).tofile(b)
    jobs.append((name, "rcm", a, b, n, ("--device",), z[f"{name}/rcm"]))
    jobs.append((name, "degree_desc", a, b, n, (), z[f"{name}/degree_desc"]))
    res, thr, grp = meta[name]["gray"][0]
    jobs.append((name, "gray", a, b, n, (str(res), str(thr), str(grp)), z[f"{name}/gray_{res}_{thr}_{grp}"]))
bad = []
t0 = time.time()
for i in range(runs):
    name, kind, a, b, n, extra, want = jobs[i % len(jobs)]
    out = os.path.join(tmp, "out.bin")
    try:
        p = subprocess.run([cli, kind, a, b, out, str(n), str(n), *extra], capture_output=True, text=True, timeout=60)
    except subprocess.TimeoutExpired:
        bad.append((i, name, kind, "TIMEOUT")); continue
    if p.returncode != 0:
        bad.append((i, name, kind, f"rc={p.returncode} {p.stderr[-300:]}")); continue
    got = np.fromfile(out, np.int32)
    if not np.array_equal(got, want):
        bad.append((i, name, kind, f"WRONG OUTPUT ({int((got != want).sum())} of {len(want)} differ)"))
print(json.dumps({"runs": runs, "seconds": round(time.time() - t0, 1), "bad": bad}, indent=1))
