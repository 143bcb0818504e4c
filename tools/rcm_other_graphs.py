#!/usr/bin/env python3
"""RCM time on inputs that are not the bench matrix (a band, a grid, a sparse random graph, many components): the
round's switches must not cost them anything.  Compare with SBX_RCM_UBFS_CHAIN=0 SBX_RCM_CC_OVERLAP=0 SBX_RCM_SPLIT_EXPAND=0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sparsebase_amd import ops, synth
cases = {"band +-64, n=1M": synth.banded_symmetric(1 << 20, 64, per_row=16, seed=1),
         "grid 1000x1000": synth.grid_graph(1000, 1000, shuffle_seed=4),
         "random avg 6, n=2M, 2 blocks": synth.random_symmetric_graph(2_000_000, avg_deg=6, seed=2, n_blocks=2, isolated_frac=0.1),
         "rmat 20 x 16": synth.rmat_symmetric(20, 16, seed=3)}
for name, (rp, col) in cases.items():
    a, b = torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda()
    for _ in range(2): ops.rcm_reorder(a, b)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): ops.rcm_reorder(a, b)
    torch.cuda.synchronize(); print("%-32s %8.3f ms" % (name, (time.perf_counter() - t) / 5 * 1e3), flush=True)
