#!/usr/bin/env python3
"""Differential fuzz of sbx_gray_row_keys against the oracle: random shapes, resolutions 1..64 (any block width),
thresholds 0..19; prints the first mismatching cases in detail.  usage: python tools/gray_fuzz.py [rounds] [seed]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from orc import Oracle
from sparsebase_amd import capi
if os.environ.get("SBX_PROBE_LIB"):  # a variant built by tools/build_variant.py
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
from sparsebase_amd import ops

def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()

def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    o = Oracle()
    bad = done = 0
    for i in range(rounds):
        shape = int(g.integers(5))
        n = int(g.integers(1, [40, 400, 4000, 200, 3][shape] + 1))
        res = int(g.integers(1, 65))
        m = res * int(g.integers(1, [4, 40, 400, 4000, 100][shape] + 1))
        lens = g.integers(0, min(m, [12, 40, 90, 300, 3000][int(g.integers(5))]) + 1, n)
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        col = np.concatenate([np.sort(g.choice(m, int(l), replace=False)) for l in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
        if len(col) == 0:
            continue
        thr = int(g.integers(0, 20))
        try:
            want = o.gray_row_keys(rp, col, m, res, thr)
        except ValueError:
            continue
        deg, key, counts = ops.gray_row_keys(m, dev(rp), dev(col), res, thr)
        done += 1
        kd, kk = deg.cpu().numpy(), key.cpu().numpy().view(np.uint64)
        ok_d, ok_k = np.array_equal(kd, want[0]), np.array_equal(kk, want[1].view(np.uint64))
        ok_c = list(counts) == list(np.asarray(want[2]).tolist())
        if not (ok_d and ok_k and ok_c):
            bad += 1
            if bad <= 6:
                rows = np.nonzero(kk != want[1].view(np.uint64))[0]
                print(f"MISMATCH round {i}: n={n} m={m} res={res} width={m // res} thr={thr} nnz={len(col)} deg_ok={ok_d} key_ok={ok_k} counts_ok={ok_c}"
                      f" got_counts={list(counts)} want_counts={np.asarray(want[2]).tolist()} bad_rows={rows[:8].tolist()} of {len(rows)}")
                for r in rows[:3]:
                    print(f"   row {r}: len={int(lens[r])} got={int(kk[r]):#x} want={int(want[1].view(np.uint64)[r]):#x} cols={col[rp[r]:rp[r+1]][:12].tolist()}")
    print(f"gray fuzz: {done} cases, {bad} mismatches")
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
