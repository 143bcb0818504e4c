#!/usr/bin/env python3
"""RCM on the bench matrix with a variant library (SBX_PROBE_LIB=<name>, tools/build_variant.py): wall time and the
per-kernel-group times, for A/B comparisons on one box."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebase_amd import capi
if os.environ.get("SBX_PROBE_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
from sparsebase_amd import ops, synth
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
order = torch.empty(rp.numel() - 1, dtype=torch.int32, device="cuda")
for _ in range(3): ops.rcm_reorder(rp, col, out=order)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): ops.rcm_reorder(rp, col, out=order)
torch.cuda.synchronize()
wall = (time.perf_counter() - t) / 20 * 1e3
ops.profile_enable(True)
for _ in range(10): ops.rcm_reorder(rp, col, out=order)
torch.cuda.synchronize()
rep = ops.profile_report(); ops.profile_enable(False)
print(os.environ.get("SBX_PROBE_LIB", "product"), "rcm %.3f ms |" % wall, " ".join(f"{k} {v[0] / 10:.3f}" for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0])[:8]))
