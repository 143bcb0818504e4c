#!/usr/bin/env python3
"""Level sizes of the BFS sweeps RCM runs on the bench matrix (diagnostic): sizes from the smallest non-isolated vertex."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparsebase_amd import synth
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
n = rp.numel() - 1
deg = (rp[1:] - rp[:-1]).long()
rpl = rp.long()
def sweep(root):
    visited = torch.zeros(n, dtype=torch.bool, device="cuda")
    f = torch.tensor([root], device="cuda")
    visited[f] = True
    sizes = []
    last = f
    while f.numel():
        sizes.append(f.numel())
        last = f
        d = deg[f]
        starts = torch.repeat_interleave(rpl[f], d)
        offs = torch.arange(int(d.sum()), device="cuda") - torch.repeat_interleave(torch.cumsum(d, 0) - d, d)
        nb = col[starts + offs].long()
        nb = nb[~visited[nb]]
        f = torch.unique(nb)
        visited[f] = True
    return sizes, last
root = int(torch.nonzero(deg > 0)[0])
for i in range(3):
    sizes, last = sweep(root)
    print("sweep", i, "root", root, "levels", sizes)
    root = int(last[torch.argmin(deg[last])])
