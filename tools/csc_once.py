import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from sparsebase_amd import ops, synth
row, col, val = synth.uniform_random_coo_torch(1 << 20, 1 << 20, 10_000_000, seed=3)
n = m = 1 << 20
rp, cc, vv = ops.coo_to_csr(n, m, row, col, val, rows_sorted=True)
for _ in range(3):
    ops.csr_to_csc(n, m, rp, cc, vv)
torch.cuda.synchronize()
