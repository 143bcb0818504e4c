#!/usr/bin/env python3
"""Permute2D on the bench matrix, random and RCM order: time of 20 warm calls each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparsebase_amd import capi
if os.environ.get("SBX_PROBE_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
from sparsebase_amd import ops, synth
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
n, nnz = rp.numel() - 1, col.numel()
val = torch.rand(nnz, device="cuda")
out = (torch.empty_like(rp), torch.empty_like(col), torch.empty_like(val))
orders = {"random": torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.int32),
          "rcm": ops.rcm_reorder(rp, col)}
for name, perm in orders.items():
    for _ in range(3): ops.permute_csr(n, n, rp, col, val, perm, perm, out=out)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): ops.permute_csr(n, n, rp, col, val, perm, perm, out=out)
    torch.cuda.synchronize(); print("permute2d %s %.3f ms" % (name, (time.perf_counter() - t) / 20 * 1e3), flush=True)
