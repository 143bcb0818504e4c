#!/usr/bin/env python3
"""RCM on deep, narrow graphs (banded / grid): levels are tiny, launch latency dominates."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebase_amd import ops, synth
res = {}
for name, (n, w, per_row) in {"banded_1M_w64": (1 << 20, 64, 12), "banded_4M_w64": (1 << 22, 64, 12)}.items():
    rp, col = synth.banded_symmetric_torch(n, w, per_row=per_row, seed=2)
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    ops.rcm_reorder(rp, col, out=out); torch.cuda.synchronize()
    t = time.perf_counter(); _, st = ops.rcm_reorder(rp, col, out=out, return_stats=True); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    res[name] = dict(n=n, nnz=col.numel(), s=round(dt, 4), mrows_s=round(n / dt / 1e6, 2), levels=st["bfs_levels"], sweeps=st["bfs_sweeps"])
print(json.dumps(res))
