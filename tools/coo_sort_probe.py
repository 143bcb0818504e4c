#!/usr/bin/env python3
"""COO constructor sort of shuffled input: C2B (10 M uniform entries, n = 2^20) and C3 (the bench matrix's 105 M entries,
shuffled).  SBX_COO_SORT_HYBRID=0 gives the plain LSD sort.  COO_PROBE_ONLY=c2b|c3 limits it (kernel traces)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparsebase_amd import capi, ops, synth
if os.environ.get("SBX_PROBE_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
only = os.environ.get("COO_PROBE_ONLY", "")
def bench(name, n, m, row, col, val, reps=6):
    r, c, v = row.clone(), col.clone(), val.clone()
    ts = []
    for _ in range(reps):
        r.copy_(row); c.copy_(col); v.copy_(val); torch.cuda.synchronize(); t = time.perf_counter()
        ops.coo_sort_(n, m, r, c, v); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(name, "%.3f ms" % (sorted(ts)[len(ts) // 2] * 1e3), flush=True)
if only in ("", "c2b"):
    n = 1 << 20
    row, col, val = synth.uniform_random_coo_torch(n, n, 10_000_000, seed=3, shuffled=True)
    bench("C2B", n, n, row, col, val)
if only in ("", "c3"):
    rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
    n = rp.numel() - 1
    rows = torch.repeat_interleave(torch.arange(n, device="cuda", dtype=torch.int32), (rp[1:] - rp[:-1]).long())
    p = torch.randperm(col.numel(), device="cuda")
    val = torch.rand(col.numel(), device="cuda")
    bench("C3 shuffled", n, n, rows[p].contiguous(), col[p].contiguous(), val[p].contiguous(), reps=4)
