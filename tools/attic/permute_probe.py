#!/usr/bin/env python3
"""Times sbx_permute_csr per kernel group (diagnostic): random Permute2D, row-wise only, RCM order."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebase_amd import ops, synth
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
rp, col = synth.rmat_symmetric_torch(scale, 13, seed=1)
n, nnz = rp.numel() - 1, col.numel()
val = torch.arange(nnz, device="cuda", dtype=torch.float32)
perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.int32)
out = (torch.empty_like(rp), torch.empty_like(col), torch.empty_like(val))
alg = 16 * nnz + 12 * n + 8
cases = {"random_2d": (perm, perm), "random_rowwise": (perm, None)}
if "--rcm" in sys.argv:
    order = ops.rcm_reorder(rp, col)
    cases["rcm_2d"] = (order, order)
res = {}
for name, (ro, co) in cases.items():
    for _ in range(2): ops.permute_csr(n, n, rp, col, val, ro, co, out=out)
    torch.cuda.synchronize()
    ops.profile_enable(True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): ops.permute_csr(n, n, rp, col, val, ro, co, out=out)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    rep = ops.profile_report(); ops.profile_enable(False)
    res[name] = dict(ms=round(ms, 3), alg_gbs=round(alg / ms / 1e6, 1), frac=round(alg / ms / 1e6 / 8000, 4),
                     kernels={k: round(v[0] / 5, 3) for k, v in rep.items()})
print(os.environ.get("SBX_DEBUG_TILE_MODE", "0"), json.dumps(res, indent=1))
