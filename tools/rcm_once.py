#!/usr/bin/env python3
"""RCM on the bench matrix: time of 10 warm calls (and the target of rocprofv3 --kernel-trace with RCM_ONCE_CALLS=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparsebase_amd import capi, ops, synth
if os.environ.get("SBX_PROBE_LIB"):  # a variant built by tools/build_variant.py
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
out = torch.empty(rp.numel() - 1, dtype=torch.int32, device="cuda")
calls = int(os.environ.get("RCM_ONCE_CALLS", "10"))
for _ in range(3): ops.rcm_reorder(rp, col, out=out)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(calls): ops.rcm_reorder(rp, col, out=out)
torch.cuda.synchronize(); print("rcm %.3f ms" % ((time.perf_counter() - t) / calls * 1e3), flush=True)
