#!/usr/bin/env python3
"""Degenerate inputs through the 64-bit entry points (empty matrices, one entry, empty rows only, nr = 0 shards)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from orc import Oracle
from sparsebase_amd import ops
orc = Oracle()
d = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
h = lambda t: None if t is None else t.cpu().numpy()
def same(got, want):
    for g, w in zip(got, want):
        assert (g is None) == (w is None)
        if g is not None: assert np.array_equal(h(g) if torch.is_tensor(g) else g, w), (h(g), w)
z = np.zeros(0, np.int64)
for n, m in ((0, 0), (5, 7), (1, 1)):
    rp = np.zeros(n + 1, np.int64)
    order = np.arange(n, dtype=np.int64)[::-1].copy()
    co = np.arange(m, dtype=np.int64)[::-1].copy()
    for v in (None, np.zeros(0, np.float64), np.zeros(0, np.float32)):
        if n:
            same(ops.permute_csr(n, m, d(rp), d(z), d(v), d(order), d(co)), orc.permute_csr(rp, z, v, order, co, m=m))
            same(ops.permute_csr(n, m, d(rp), d(z), d(v), d(order), None), orc.permute_csr(rp, z, v, order, None, m=m))
        same(ops.coo_to_csc(n, m, d(z), d(z), d(v)), orc.coo_to_csc(n, m, z, z, v))
        same(ops.csr_to_csc(n, m, d(rp), d(z), d(v)), orc.csr_to_csc(m, rp, z, v))
        r, c, vv = d(z.copy()), d(z.copy()), d(None if v is None else v.copy())
        ops.coo_sort_(n, m, r, c, vv)
# one entry, duplicates, a shard of no rows
rp = np.array([0, 0, 3, 3], np.int64); col = np.array([2, 0, 2], np.int64); val = np.array([3.0, 1.0, 2.0])
order = np.array([2, 0, 1], np.int64); co = np.array([1, 2, 0], np.int64)
same(ops.permute_csr(3, 3, d(rp), d(col), d(val), d(order), d(co)), orc.permute_csr(rp, col, val, order, co, m=3))
srp, sc, sv = ops.permute_csr_rows(3, 3, d(rp), d(col), d(val), d(order), d(co), 1, 1)
assert h(srp).tolist() == [0] and sc.numel() == 0
r, c, v = d(np.array([2, 0, 2, 0], np.int64)), d(np.array([1, 5, 0, 5], np.int64)), d(np.array([4.0, 2.0, 3.0, 1.0]))
ops.coo_sort_(3, 6, r, c, v)
assert h(r).tolist() == [0, 0, 2, 2] and h(c).tolist() == [5, 5, 0, 1], (h(r), h(c))
same(ops.coo_to_csc(3, 6, d(np.array([2, 0, 2], np.int64)), d(np.array([1, 5, 0], np.int64)), d(np.array([4.0, 2.0, 3.0]))),
     orc.coo_to_csc(3, 6, np.array([2, 0, 2], np.int64), np.array([1, 5, 0], np.int64), np.array([4.0, 2.0, 3.0])))
print("int64 degenerate ok")
