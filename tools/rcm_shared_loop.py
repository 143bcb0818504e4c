#!/usr/bin/env python3
"""RCM against the oracle in a loop (argv[1] rounds over four graphs), for several of these at once on one GPU
(tools/rcm_shared_loop.sh): grid barriers give up now and then, sweeps are redone — the order must not change.
SBX_DEBUG_GB_BACKOFF=0 makes every call try the persistent kernels again."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from sparsebase_amd import capi
if os.environ.get("SBX_PROBE_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
from orc import Oracle
from sparsebase_amd import ops, synth
orc = Oracle()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
cases = [synth.rmat_symmetric(16, 16, seed=1), synth.rmat_symmetric(14, 8, seed=1), synth.rmat_symmetric(17, 8, seed=5),
         synth.random_symmetric_graph(40000, avg_deg=4, seed=2, n_blocks=3, isolated_frac=0.1)]
dev = [(torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda(), orc.rcm_reorder(rp, col)) for rp, col in cases]
bad = exc = giveups = 0
for r in range(rounds):
    for i, (rp, col, want) in enumerate(dev):
        try:
            got, st = ops.rcm_reorder(rp, col, return_stats=True)
            if not np.array_equal(got.cpu().numpy(), want): bad += 1; print("MISMATCH round", r, "case", i, flush=True)
        except Exception as e:
            exc += 1; print("EXC round", r, "case", i, str(e)[:100], flush=True)
if "--digests" in sys.argv:  # what the fenced-build comparison reads: one digest per graph, of the last round's order
    import hashlib
    for i, (rp, col, want) in enumerate(dev):
        print("digest", i, hashlib.sha256(ops.rcm_reorder(rp, col).cpu().numpy().tobytes()).hexdigest(), flush=True)
print("loop: rounds", rounds, "mismatches", bad, "exceptions", exc)
sys.exit(1 if bad or exc else 0)
