"""RCM against the oracle on RMAT graphs of scale 14 - 18, three calls each (tools/rcm_shared_gpu.sh runs several of these
at once on one GPU)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
from sparsebase_amd import capi, ops, synth
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
if os.environ.get("SBX_PROBE_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
from orc import Oracle
orc = Oracle()
bad = 0
CASES = ((14, 8, 1), (16, 13, 1), (16, 16, 3), (17, 8, 5), (18, 13, 1))
if os.environ.get("RCM_MODES_SMALL"):
    CASES = CASES[:3]
for scale, ef, seed in CASES:
    rp, col = synth.rmat_symmetric(scale, ef, seed=seed)
    want = orc.rcm_reorder(rp, col)
    for rep in range(3):
        try:
            got = ops.rcm_reorder(torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda()).cpu().numpy()
            ok = np.array_equal(got, want)
        except Exception as e:
            ok = False; print("EXC", scale, ef, seed, str(e)[:60] + " ... " + str(e)[-12:])
        if not ok: bad += 1; print("MISMATCH", scale, ef, seed, rep)
print("modes", os.environ.get("SBX_RCM_UNORDERED"), os.environ.get("SBX_DEBUG_GB_SPINS"), os.environ.get("SBX_RCM_COUNT_SORT"), "bad", bad)
sys.exit(1 if bad else 0)
