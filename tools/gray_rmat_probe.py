#!/usr/bin/env python3
"""Gray key stage on the power-law bench matrix (rows of every length: the mixed path) and on the banded C5 matrices."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparsebase_amd import capi, ops, synth
if os.environ.get("SBX_PROBE_LIB"):  # a variant built by tools/build_variant.py
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
def bench(name, n, rp, col, res, thr):
    for _ in range(3): ops.gray_row_keys(n, rp, col, res, thr)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): ops.gray_row_keys(n, rp, col, res, thr)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 10 * 1e3
    nnz = col.numel()
    print(f"{name}: {ms:.3f} ms  {(4 * nnz + 16 * n) / (ms * 1e-3) / 8e12:.3f} of 8 TB/s", flush=True)
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
n = rp.numel() - 1
bench("RMAT scale 22 (32, 10)", n, rp, col, 32, 10)
bench("RMAT scale 22 (16, 20)", n, rp, col, 16, 20)
for hb in (64, n // 16):
    rp, col = synth.banded_symmetric_torch(n, hb, per_row=12, seed=2)
    bench(f"banded +-{hb} (32, 10)", n, rp, col, 32, 10)
