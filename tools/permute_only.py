#!/usr/bin/env python3
"""Runs Permute2D (random order, or --rcm) a few times on the bench matrix: the target of rocprofv3 runs."""
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
val = torch.arange(nnz, device="cuda", dtype=torch.float32)
if "--rcm" in sys.argv:
    perm = ops.rcm_reorder(rp, col)
else:
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.int32)
if "--prerelabel" in sys.argv:
    # sort-only ablation (with SBX_PERMUTE_FORCE_RADIX=4: the kernels skip the relabel gathers): the columns arrive relabelled,
    # so the sort sees the keys of the real call and no gather is issued
    col = perm[col.long()].contiguous()
out = (torch.empty_like(rp), torch.empty_like(col), torch.empty_like(val))
for _ in range(6):
    ops.permute_csr(n, n, rp, col, val, perm, perm, out=out)
torch.cuda.synchronize()
