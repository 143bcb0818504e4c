#!/usr/bin/env python3
"""One warm Permute2D (random order) on the bench matrix with a variant library (diagnostic builds print to stderr)."""
import os, sys
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
perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.int32)
if "--rcm" in sys.argv:
    perm = ops.rcm_reorder(rp, col)
out = (torch.empty_like(rp), torch.empty_like(col), torch.empty_like(val))
for i in range(int(os.environ.get("REPS", "3"))):
    print("--- call", i, file=sys.stderr, flush=True)
    ops.permute_csr(n, n, rp, col, val, perm, perm, out=out)
    torch.cuda.synchronize()
