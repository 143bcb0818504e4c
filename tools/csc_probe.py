#!/usr/bin/env python3
"""CSR->CSC / COO->CSC timing: C2-sized uniform matrix and the 105M-nnz RMAT."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebase_amd import ops, synth

def timed(f, reps=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

out = {}
row, col, val = synth.uniform_random_coo_torch(1 << 20, 1 << 20, 10_000_000, seed=3)
n = m = 1 << 20
rp, cc, vv = ops.coo_to_csr(n, m, row, col, val, rows_sorted=True)
for name, (n_, m_, rp_, cc_, vv_) in {"c2_uniform_10m": (n, m, rp, cc, vv)}.items():
    nnz = cc_.numel()
    alg = 20 * nnz + 4 * (n_ + 1) + 4 * (m_ + 1)
    ms = timed(lambda: ops.csr_to_csc(n_, m_, rp_, cc_, vv_))
    ms2 = timed(lambda: ops.coo_to_csc(n_, m_, row, col, val))
    out[name] = dict(nnz=nnz, csr_to_csc_ms=round(ms, 3), coo_to_csc_ms=round(ms2, 3), alg_gbs=round(alg / ms / 1e6, 1))
rp, cc = synth.rmat_symmetric_torch(22, 13, seed=1)
n = rp.numel() - 1
vv = torch.arange(cc.numel(), device="cuda", dtype=torch.float32)
nnz = cc.numel()
ops.profile_enable(True)
ms = timed(lambda: ops.csr_to_csc(n, n, rp, cc, vv))
rep = ops.profile_report(); ops.profile_enable(False)
out["rmat22"] = dict(nnz=nnz, csr_to_csc_ms=round(ms, 3), alg_gbs=round((20 * nnz + 8 * n) / ms / 1e6, 1),
                     kernels={k: round(v[0] / 7, 3) for k, v in rep.items()})
print(json.dumps(out, indent=1))
