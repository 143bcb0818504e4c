#!/usr/bin/env python3
"""Permute2D on the bench matrix with a variant library (SBX_PROBE_LIB=<name>): float values, pattern only, doubles."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebase_amd import capi
if os.environ.get("SBX_PROBE_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
from sparsebase_amd import ops, synth
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
n, nnz = rp.numel() - 1, col.numel()
perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.int32)
res = []
for name, val in (("f32", torch.ones(nnz, device="cuda")), ("pattern", None), ("f64", torch.ones(nnz, device="cuda", dtype=torch.float64))):
    out = (torch.empty_like(rp), torch.empty_like(col), None if val is None else torch.empty_like(val))
    for _ in range(3): ops.permute_csr(n, n, rp, col, val, perm, perm, out=out)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): ops.permute_csr(n, n, rp, col, val, perm, perm, out=out)
    torch.cuda.synchronize(); res.append("%s %.3f ms" % (name, (time.perf_counter() - t) / 20 * 1e3))
print(os.environ.get("SBX_PROBE_LIB", "product"), " | ".join(res))
