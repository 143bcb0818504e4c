#!/usr/bin/env python3
"""RCM against the oracle with grid barriers that give up after SBX_DEBUG_GB_SPINS polls (set by the caller; read once per
process): a few graphs, rotated by argv[1], five rounds.  SBX_PROBE_LIB=<name> runs a library variant."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from sparsebase_amd import capi
if os.environ.get("SBX_PROBE_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
from orc import Oracle
from sparsebase_amd import ops, synth
orc = Oracle()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
cases = [synth.rmat_symmetric(16, 8, seed=3), synth.rmat_symmetric(17, 6, seed=11), synth.rmat_symmetric(15, 12, seed=5),
         synth.random_symmetric_graph(60000, avg_deg=5, seed=2, n_blocks=2, isolated_frac=0.1)]
k = int(sys.argv[1]) % len(cases) if len(sys.argv) > 1 else 0
cases = cases[k:] + cases[:k]
bad = 0
for i, (rp, col) in enumerate(cases * 5):
    try:
        ok = np.array_equal(ops.rcm_reorder(d(rp), d(col)).cpu().numpy(), orc.rcm_reorder(rp, col))
    except Exception as e:
        ok = False; print("exception:", e)
    bad += not ok
print("spins", os.environ.get("SBX_DEBUG_GB_SPINS"), "rotation", k, "mismatches", bad)
