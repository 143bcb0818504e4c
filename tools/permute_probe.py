#!/usr/bin/env python3
"""Times sbx_permute_csr per kernel group (diagnostic)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebase_amd import ops, synth
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
rp, col = synth.rmat_symmetric_torch(scale, 13, seed=1)
n, nnz = rp.numel() - 1, col.numel()
val = torch.arange(nnz, device="cuda", dtype=torch.float32)
perm = torch.randperm(n, device="cuda").to(torch.int32)
out = (torch.empty_like(rp), torch.empty_like(col), torch.empty_like(val))
for _ in range(2): ops.permute_csr(n, n, rp, col, val, perm, perm, out=out)
torch.cuda.synchronize()
ops.profile_enable(True)
for _ in range(5): ops.permute_csr(n, n, rp, col, val, perm, perm, out=out)
rep = ops.profile_report()
print(os.environ.get("SBX_DEBUG_TILE_MODE", "0"), json.dumps({k: round(v[0] / 5, 3) for k, v in rep.items()}))
