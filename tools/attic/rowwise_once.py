import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from sparsebase_amd import ops, synth
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
n, nnz = rp.numel() - 1, col.numel()
val = torch.arange(nnz, device="cuda", dtype=torch.float32)
perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.int32)
for _ in range(4):
    ops.permute_csr(n, n, rp, col, val, perm, None)
torch.cuda.synchronize()
