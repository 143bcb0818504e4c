#!/usr/bin/env python3
"""Config C5: GrayReorder device stage on ~100M-nnz banded matrices (and the RMAT one): time + kernel split."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebase_amd import capi
if os.environ.get("SBX_PROBE_LIB"):  # a variant built by tools/build_variant.py
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
from sparsebase_amd import ops, synth
n = 1 << 22
cases = {"banded_w64": lambda: synth.banded_symmetric_torch(n, 64, per_row=12, seed=2),
         "banded_w_m16": lambda: synth.banded_symmetric_torch(n, n // 16, per_row=12, seed=2),
         "rmat22": lambda: synth.rmat_symmetric_torch(22, 13, seed=1)}
if os.environ.get("C5_ONLY_BANDED"):
    cases.pop("rmat22")
out = {}
for name, make in cases.items():
    rp, col = make()
    nnz = col.numel()
    for _ in range(2): ops.gray_row_keys(n, rp, col, 32, 10)
    torch.cuda.synchronize()
    ops.profile_enable(True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): ops.gray_row_keys(n, rp, col, 32, 10)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    rep = ops.profile_report(); ops.profile_enable(False)
    alg = 4 * nnz + 8 * n + 4
    out[name] = dict(nnz=nnz, ms=round(ms, 3), mrows_s=round(n / ms / 1e3, 1), alg_gbs=round(alg / ms / 1e6, 1),
                     frac_8tbs=round(alg / ms / 1e6 / 8000, 4), kernels={k: round(v[0] / 5, 3) for k, v in rep.items()})
    del rp, col
print(json.dumps(out, indent=1))
