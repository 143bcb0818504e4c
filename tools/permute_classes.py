#!/usr/bin/env python3
"""Entries of the bench matrix per row-length class of the permute (row lengths do not change under a permutation)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparsebase_amd import synth
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
ln = (rp[1:] - rp[:-1]).to(torch.int64)
edges = [0, 128, 256, 512, 1024, 2048, 4096, 8192, 1 << 40]
names = ["tile(<=128)", "c256", "c512", "c1024", "c2048", "c4096", "c8192", "long"]
for lo, hi, nm in zip(edges[:-1], edges[1:], names):
    m = (ln > lo) & (ln <= hi)
    print(f"{nm:12s} rows {int(m.sum()):9d} entries {int(ln[m].sum()):11d}")
# slot utilisation of the capacity classes: entries / (rows x capacity)
for lo, hi, nm in zip(edges[1:-2], edges[2:-1], names[1:-1]):
    m = (ln > lo) & (ln <= hi)
    r, e = int(m.sum()), int(ln[m].sum())
    print(f"{nm:12s} utilisation {e / max(1, r * hi):.2f}  mean length {e / max(1, r):.0f}")
