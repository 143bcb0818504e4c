#!/usr/bin/env python3
"""Differential fuzz of Permute2D on matrices with rows above the one-workgroup capacity (the segment path of
sbx_permute.hip) against the oracle: few rows, lengths up to 300 K, uniform / clustered / multi-scale column sets,
random / monotone / block-scrambling column maps, duplicates, every value width.
usage: python tools/permute_long_fuzz.py [rounds] [seed]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from orc import Oracle
from sparsebase_amd import ops

def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()

def host(t):
    return None if t is None else t.cpu().numpy()

def columns(g, m, l, kind, dups):
    if l == 0:
        return np.zeros(0, np.int64)
    if kind == 0:      # uniform
        c = g.integers(0, m, l) if dups else g.choice(m, min(l, m), replace=False)
    elif kind == 1:    # one tight cluster + outliers
        w = max(min(m, 2 * l), 1)
        b = int(g.integers(0, m - w + 1))
        c = b + (g.integers(0, w, l) if dups else g.choice(w, min(l, w), replace=False))
        k = int(g.integers(0, 4))
        if k and l > k:
            c[:k] = g.integers(0, m, k)
    else:              # clusters at several scales
        parts, left = [], l
        while left > 0:
            k = int(min(left, max(1, g.integers(1, max(2, l // 3)))))
            w = int(min(m, max(k, g.integers(k, max(k + 1, 8 * k)))))
            b = int(g.integers(0, m - w + 1))
            parts.append(b + (g.integers(0, w, k) if dups else g.choice(w, k, replace=False)))
            left -= k
        c = np.concatenate(parts)
        if not dups:
            c = np.unique(c)
    return np.sort(c)

def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    o = Oracle()
    bad = 0
    for i in range(rounds):
        m = int(2 ** g.integers(14, 23)) + int(g.integers(0, 1000))
        n = int(g.integers(1, 24))
        dups = bool(g.integers(0, 4) == 0)
        lens = [int(x) for x in np.minimum(m if not dups else 10 ** 9, g.choice([0, 3, 200, 5000, 8192, 8193, 9000, 20000, 70000, 300000], n))]
        lens = [int(l * g.uniform(0.7, 1.0)) if l > 10000 else l for l in lens]
        cols = [columns(g, m, l, int(g.integers(0, 3)), dups) for l in lens]
        lens = [len(c) for c in cols]
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        col = np.concatenate(cols + [np.zeros(0, np.int64)]).astype(np.int32)
        kind = int(g.integers(0, 4))
        if kind == 0:
            co = g.permutation(m)
        elif kind == 1:
            co = np.arange(m)
        elif kind == 2:   # scramble inside blocks of 2^k columns
            k = int(g.integers(2, 16))
            base = np.arange(m)
            co = (base & ~((1 << k) - 1)) | g.permutation(1 << k)[base & ((1 << k) - 1)]
            co = np.argsort(np.argsort(co, kind="stable"), kind="stable") if co.max() >= m else co
        else:             # half monotone, half scrambled
            co = np.arange(m)
            co[m // 2:] = m // 2 + g.permutation(m - m // 2)
        co = co.astype(np.int32)
        ro = g.permutation(n).astype(np.int32)
        vk = int(g.integers(0, 4))
        val = [None, g.integers(-4, 4, len(col)).astype(np.int32), g.random(len(col)).astype(np.float32), g.integers(-4, 4, len(col)).astype(np.float64)][vk]
        want = o.permute_csr(rp, col, val, ro, co)
        got = ops.permute_csr(n, m, dev(rp), dev(col), dev(val), dev(ro), dev(co))
        ok = all((a is None and b is None) or np.array_equal(host(a), b) for a, b in zip(got, want))
        if not ok:
            bad += 1
            print(f"MISMATCH round {i}: n={n} m={m} lens={lens} dups={dups} map={kind} val={vk}", flush=True)
    print(f"permute long-row fuzz: {rounds} rounds, {bad} mismatches")
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
