#!/usr/bin/env python3
"""Differential fuzz of the C ABI against the CPU restatement (bit-exact) on the GPU box: random shapes, skewed
row lengths, empty rows, 32/64-bit indices, every value type, square and rectangular, distinct coordinates
(where the reference's unstable sorts make the result unique).  usage: python tools/fuzz_ops.py [rounds] [seed]"""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from orc import Oracle  # noqa: E402
from sparsebase_amd import ops  # noqa: E402

VDT = [None, np.int32, np.float32, np.float64, np.int64]
STATS = {}


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return None if t is None else t.cpu().numpy()


def same(a, b):
    if a is None or b is None:
        return a is None and b is None
    a, b = np.atleast_1d(np.asarray(a)), np.atleast_1d(np.asarray(b))
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def random_matrix(g, idt):
    """Distinct coordinates; returns n, m, shuffled COO and the sorted CSR."""
    shape = int(g.integers(6))
    n = int(g.integers(1, [40, 400, 4000, 40000, 200, 3][shape] + 1))
    m = n if g.random() < 0.5 else int(g.integers(1, [40, 400, 4000, 40000, 70000, 100000][shape] + 1))
    dens = [0.0, 0.5, 2.0, 8.0, 40.0, 3000.0][int(g.integers(6))]
    want = int(min(n * m, max(0, dens * n * g.random()), 400000))
    if g.random() < 0.3 and n > 4:          # power-law rows: a few rows own most entries
        w = g.pareto(1.2, n) + 1e-3
        rows = g.choice(n, size=want, p=w / w.sum())
    else:
        rows = g.integers(0, n, want)
    cols = g.integers(0, m, want)
    key = np.unique(rows.astype(np.int64) * m + cols)
    p = g.permutation(len(key))
    r, c = (key // m).astype(idt), (key % m).astype(idt)
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, r + 1, 1)
    rp = np.cumsum(rp).astype(idt)
    return n, m, r[p], c[p], rp, c.copy(), p


def values(g, vdt, k):
    if vdt is None:
        return None
    if np.issubdtype(vdt, np.integer):
        return g.integers(-1000, 1000, k).astype(vdt)
    return (g.standard_normal(k) * 10.0 ** g.integers(-3, 4)).astype(vdt)


def one_round(g, o, i):
    idt = np.int32 if g.random() < 0.75 else np.int64
    vdt = VDT[int(g.integers(len(VDT)))]
    n, m, srow, scol, rp, col, p = random_matrix(g, idt)
    nnz = len(col)
    val = values(g, vdt, nnz)                      # values in CSR (sorted) order
    sval = None if val is None else val[p]         # ... and in the shuffled COO order
    tag = f"round {i}: n={n} m={m} nnz={nnz} idx={np.dtype(idt).name} val={None if vdt is None else np.dtype(vdt).name}"
    bad = []

    def check(name, got, want):
        STATS[name] = STATS.get(name, 0) + 1
        STATS['nnz'] = STATS.get('nnz', 0) + nnz
        got = got if isinstance(got, (tuple, list)) else (got,)
        want = want if isinstance(want, (tuple, list)) else (want,)
        if len(got) != len(want) or not all(same(host(a) if torch.is_tensor(a) else a, b) for a, b in zip(got, want)):
            bad.append(name)

    # constructor sorts
    r_, c_, v_ = dev(srow), dev(scol), dev(sval)
    ops.coo_sort_(n, m, r_, c_, v_)
    check("coo_sort", (r_, c_, v_), o.coo_sort(srow, scol, sval, n=n, m=m))
    if nnz:
        q = np.concatenate([rp[j] + g.permutation(rp[j + 1] - rp[j]) for j in range(n)]) if n <= 4000 else np.arange(nnz)
        q = q.astype(np.int64)
        c2, v2 = dev(col[q]), dev(None if val is None else val[q])
        ops.csr_sort_rows_(n, m, dev(rp), c2, v2)
        check("csr_sort_rows", (c2, v2), o.csr_sort_rows(rp, col[q], None if val is None else val[q], m=m))
    # conversions
    row_sorted = np.repeat(np.arange(n, dtype=idt), np.diff(rp).astype(np.int64))
    check("coo_to_csr", ops.coo_to_csr(n, m, dev(row_sorted), dev(col), dev(val)), o.coo_to_csr(n, row_sorted, col, val, m=m))
    check("csr_to_coo", ops.csr_to_coo(n, m, dev(rp), dev(col), dev(val)), o.csr_to_coo(rp, col, val, m=m))
    if vdt is None or np.dtype(vdt).itemsize in (4, 8):
        check("csr_to_csc", ops.csr_to_csc(n, m, dev(rp), dev(col), dev(val)), o.csr_to_csc(m, rp, col, val))
        check("coo_to_csc", ops.coo_to_csc(n, m, dev(row_sorted), dev(col), dev(val)), o.coo_to_csc(n, m, row_sorted, col, val))
    # permutation apply
    ro = g.permutation(n).astype(idt) if g.random() < 0.8 else None
    co = g.permutation(m).astype(idt) if g.random() < 0.7 else None
    if ro is not None or co is not None:
        check("permute_csr", ops.permute_csr(n, m, dev(rp), dev(col), dev(val), dev(ro), dev(co)),
              o.permute_csr(rp, col, val, ro, co, m=m))
    if ro is not None:
        check("inverse_permutation", ops.inverse_permutation(dev(ro)), o.inverse_permutation(ro))
        if val is not None and n:
            arr = values(g, vdt, n)
            check("permute_array", ops.permute_array(dev(ro), dev(arr)), o.permute_array(ro, arr))
    # reorderers and features
    asc = bool(g.integers(2))
    check("degree_reorder", ops.degree_reorder(dev(rp), asc), o.degree_reorder(rp, asc))
    check("csr_degrees", ops.csr_degrees(dev(rp)), o.csr_degrees(rp))
    if n == m:
        check("csr_bandwidth", np.int64(ops.csr_bandwidth(dev(rp), dev(col))), np.int64(o.csr_bandwidth(rp, col)))
        check("csr_profile", np.int64(ops.csr_profile(dev(rp), dev(col))), np.int64(o.csr_profile(rp, col)))
    res = int(g.integers(1, 65))
    thr = int(g.integers(0, 20))
    try:
        want = o.gray_row_keys(rp, col, m, res, thr)
    except ValueError:
        want = None
    if want is not None and nnz:
        deg, key, counts = ops.gray_row_keys(m, dev(rp), dev(col), res, thr)
        check("gray_row_keys", (deg, host(key).view(np.uint64), np.array(counts, np.int64)),
              (want[0], want[1].view(np.uint64), np.asarray(want[2], np.int64)))
    if n == m and idt == np.int32 and nnz:
        # symmetric pattern for RCM
        rr = np.concatenate([row_sorted, col]).astype(np.int64)
        cc = np.concatenate([col, row_sorted]).astype(np.int64)
        k2 = np.unique(rr * n + cc)
        r2, c2 = (k2 // n).astype(idt), (k2 % n).astype(idt)
        rp2 = np.zeros(n + 1, np.int64)
        np.add.at(rp2, r2 + 1, 1)
        rp2 = np.cumsum(rp2).astype(idt)
        check("rcm_reorder", ops.rcm_reorder(dev(rp2), dev(c2)), o.rcm_reorder(rp2, c2))
    if bad:
        print("MISMATCH", tag, bad, flush=True)
    return len(bad)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    g = np.random.default_rng(seed)
    o = Oracle()
    bad = errors = 0
    for i in range(rounds):
        try:
            bad += one_round(g, o, i)
        except Exception:
            errors += 1
            print(f"EXCEPTION in round {i}", flush=True)
            traceback.print_exc()
            if errors > 5:
                break
    print("checks:", {k: v for k, v in sorted(STATS.items())})
    print(f"fuzz: {rounds} rounds, {bad} mismatching ops, {errors} exceptions")
    return 1 if (bad or errors) else 0


if __name__ == "__main__":
    sys.exit(main())
