#!/usr/bin/env python3
"""GrayReorder end to end through the C++ API (device key stage + host ordering stage, device-resident input)
next to the real reference on the host, results compared."""
import json, os, subprocess, sys, tempfile, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sparsebase_amd import synth
import orc
ref = orc.Ref() if orc.ref_available() else None
cli = os.path.join(ROOT, "sparsebase_amd", "host", "bin", "reorder_cli")
tmp = tempfile.mkdtemp()
n = 1 << 22
cases = {"banded_w64": lambda: synth.banded_symmetric_torch(n, 64, per_row=12, seed=2),
         "banded_w_m16": lambda: synth.banded_symmetric_torch(n, n // 16, per_row=12, seed=2),
         "rmat22": lambda: synth.rmat_symmetric_torch(22, 13, seed=1)}
out = {}
for name, make in cases.items():
    rp, col = (t.cpu().numpy() for t in make())
    a, b, o = (os.path.join(tmp, x) for x in ("rp.bin", "col.bin", "out.bin"))
    rp.tofile(a); col.tofile(b)
    r = {"nnz": len(col)}
    for tag, params in (("32_10_4", (32, 10, 4)),):
        p = subprocess.run([cli, "gray", a, b, o, str(n), str(n), *map(str, params), "--device", "--time"],
                           capture_output=True, text=True, timeout=600)
        lines = p.stdout.strip().splitlines()
        r["gpu_path_s"] = float(lines[-2])
        r["stages_ms(device keys, keys to host, host ordering)"] = [float(x) for x in lines[-1].split()]
        host = [l for l in p.stderr.splitlines() if l.startswith("gray host stage")]
        if host:
            r["host_stage"] = host[-1]
        got = np.fromfile(o, np.int32)
        if ref is not None:
            t = time.perf_counter(); want = ref.gray_reorder(rp, col, n, *params); r["reference_s"] = round(time.perf_counter() - t, 3)
            r["identical"] = bool(np.array_equal(got, want))
            r["speedup"] = round(r["reference_s"] / r["gpu_path_s"], 2)
    out[name] = r
print(json.dumps(out, indent=1))
