#!/usr/bin/env python3
"""DegreeReorder on the bench matrix: time of 20 warm calls (and a target for rocprofv3 --kernel-trace)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparsebase_amd import capi, ops, synth
if os.environ.get("SBX_PROBE_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
out = torch.empty(rp.numel() - 1, dtype=torch.int32, device="cuda")
for asc in (True, False):
    for _ in range(3): ops.degree_reorder(rp, asc, out=out)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): ops.degree_reorder(rp, asc, out=out)
    torch.cuda.synchronize(); print("degree asc=%s %.3f ms" % (asc, (time.perf_counter() - t) / 20 * 1e3), flush=True)
