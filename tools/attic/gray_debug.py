#!/usr/bin/env python3
"""Which rows of the Gray key stage differ from the oracle, and how long they are (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from orc import Oracle
from sparsebase_amd import capi, ops, synth
if os.environ.get("SBX_PROBE_LIB"):  # a variant built by tools/build_variant.py
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
orc = Oracle()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for seed, (res, thr) in enumerate([(32, 10), (16, 20), (64, 2), (16, 0)]):
    rp, col = synth.rmat_symmetric(12, 8, seed=seed) if seed % 2 == 0 else synth.banded_symmetric(4096, 40, 9, seed)
    n = len(rp) - 1
    deg, key, counts = ops.gray_row_keys(n, dev(rp), dev(col), res, thr)
    wdeg, wkey, wcounts = orc.gray_row_keys(rp, col, n, res, thr)
    d = np.diff(rp)
    bad_deg = np.nonzero(deg.cpu().numpy() != wdeg)[0]
    bad_key = np.nonzero(key.cpu().numpy().view(np.uint64) != wkey)[0]
    print(seed, res, thr, "max len", d.max(), "rows>64", (d > 64).sum(), "rows>1024", (d > 1024).sum())
    print("  bad deg rows", bad_deg[:10], d[bad_deg[:10]], "bad key rows", len(bad_key), bad_key[:10], d[bad_key[:10]])
    print("  counts", list(counts), wcounts.tolist())
