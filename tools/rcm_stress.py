#!/usr/bin/env python3
"""RCM stress on the GPU box: many graph families, GPU result vs the CPU restatement (bit-exact), with the number of
BFS sweeps the speculative policy needed.  usage: python tools/rcm_stress.py [rounds]"""
import collections
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from orc import Oracle  # noqa: E402
from sparsebase_amd import ops, synth  # noqa: E402


def sym_from_edges(n, u, v):
    r = np.concatenate([u, v]).astype(np.int64)
    c = np.concatenate([v, u]).astype(np.int64)
    key = np.unique(r * n + c)
    r, c = (key // n).astype(np.int32), (key % n).astype(np.int32)
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, r + 1, 1)
    return np.cumsum(rp).astype(np.int32), c


def families(g, i):
    kind = i % 7
    if kind == 0:      # sparse random graph near the connectivity threshold: long peripheral searches
        n = int(g.integers(200, 20000))
        e = int(n * g.uniform(0.6, 2.5))
        return "gnm", sym_from_edges(n, g.integers(0, n, e), g.integers(0, n, e))
    if kind == 1:      # random tree (eccentricities vary a lot between candidates)
        n = int(g.integers(100, 30000))
        v = np.arange(1, n)
        u = (g.random(n - 1) * v).astype(np.int64)
        p = g.permutation(n)
        return "tree", sym_from_edges(n, p[u], p[v])
    if kind == 2:      # caterpillar / broom: a long path with random pendant bushes
        n = int(g.integers(100, 20000))
        spine = int(n * g.uniform(0.2, 0.9))
        u = list(range(spine - 1)) + list(g.integers(0, spine, n - spine))
        v = list(range(1, spine)) + list(range(spine, n))
        p = g.permutation(n)
        return "broom", sym_from_edges(n, p[np.array(u)], p[np.array(v)])
    if kind == 3:      # shuffled 2-D grid with a few random chords
        a, b = int(g.integers(5, 120)), int(g.integers(5, 120))
        n = a * b
        idx = np.arange(n).reshape(a, b)
        u = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel(), g.integers(0, n, 3)])
        v = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel(), g.integers(0, n, 3)])
        p = g.permutation(n)
        return "grid", sym_from_edges(n, p[u], p[v])
    if kind == 4:      # power law
        scale = int(g.integers(8, 15))
        rp, col = synth.rmat_symmetric(scale, int(g.integers(2, 12)), seed=int(g.integers(1 << 30)))
        return "rmat", (rp.astype(np.int32), col.astype(np.int32))
    if kind == 5:      # several mid-size components of different shapes
        parts, off, us, vs = int(g.integers(2, 8)), 0, [], []
        for _ in range(parts):
            k = int(g.integers(70, 3000))
            e = int(k * g.uniform(1.0, 3.0))
            us.append(off + g.integers(0, k, e))
            vs.append(off + g.integers(0, k, e))
            us.append(off + np.arange(k - 1))
            vs.append(off + np.arange(1, k))
            off += k
        p = g.permutation(off)
        return "multi", sym_from_edges(off, p[np.concatenate(us)], p[np.concatenate(vs)])
    n = int(g.integers(300, 20000))       # random band
    w = int(g.integers(1, 40))
    u = g.integers(0, n, n * 3)
    v = np.clip(u + g.integers(-w, w + 1, n * 3), 0, n - 1)
    return "band", sym_from_edges(n, u, v)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 140
    oracle = Oracle()
    g = np.random.default_rng(20261003)
    hist = collections.defaultdict(collections.Counter)
    bad = 0
    for i in range(rounds):
        name, (rp, col) = families(g, i)
        want = oracle.rcm_reorder(rp, col)
        d_rp, d_col = torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda()
        got, stats = ops.rcm_reorder(d_rp, d_col, return_stats=True)
        same = np.array_equal(got.cpu().numpy(), want)
        bad += not same
        hist[name][int(stats["bfs_sweeps"])] += 1
        if not same:
            print("MISMATCH", name, i, len(rp) - 1, len(col))
    print(json.dumps({k: dict(sorted(v.items())) for k, v in hist.items()}))
    print("mismatches:", bad, "of", rounds)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
