#!/usr/bin/env python3
"""Prints the in-kernel phase stamps of k_permute_tile (run with SBX_DEBUG_TILE_STOP=9)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparsebase_amd import ops, synth
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
n, nnz = rp.numel() - 1, col.numel()
val = torch.arange(nnz, device="cuda", dtype=torch.float32)
perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.int32)
if "--rcm" in sys.argv:
    perm = ops.rcm_reorder(rp, col)
out = (torch.empty_like(rp), torch.empty_like(col), torch.empty_like(val))
for _ in range(3):
    ops.permute_csr(n, n, rp, col, val, perm, perm, out=out)
torch.cuda.synchronize()
