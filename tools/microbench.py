#!/usr/bin/env python3
"""Quick per-operation timing on one GPU (HIP events on the handle's stream). Not the scored bench."""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebase_amd import ops, synth

def timeit(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=20)
ap.add_argument("--ef", type=int, default=16)
ap.add_argument("--coo-nnz", type=int, default=10_000_000)
ap.add_argument("--skip", default="")
args = ap.parse_args()
res = {}
t = time.time()
rp, col = synth.rmat_symmetric_torch(args.scale, args.ef, seed=1)
torch.cuda.synchronize()
n, nnz = rp.numel() - 1, col.numel()
print(f"rmat scale {args.scale}: n={n} nnz={nnz} gen {time.time()-t:.1f}s", flush=True)
val = torch.arange(nnz, device="cuda", dtype=torch.float32)
perm = torch.randperm(n, device="cuda").to(torch.int32)
out = (torch.empty_like(rp), torch.empty_like(col), torch.empty_like(val))
ms = timeit(lambda: ops.permute_csr(n, n, rp, col, val, perm, perm, out=out))
alg = 16 * nnz + 12 * n + 8
res["permute2d_random"] = dict(ms=ms, mrows_s=n / ms / 1e3, alg_gbs=alg / ms / 1e6)
ms = timeit(lambda: ops.permute_csr(n, n, rp, col, val, perm, None, out=out))
res["permute_rowwise"] = dict(ms=ms, mrows_s=n / ms / 1e3, alg_gbs=alg / ms / 1e6)
coo_out = (torch.empty_like(col), torch.empty_like(col), torch.empty_like(val))
ms = timeit(lambda: ops.csr_to_coo(n, n, rp, col, val, out=coo_out))
alg = 20 * nnz + 4 * (n + 1)
res["csr_to_coo"] = dict(ms=ms, alg_gbs=alg / ms / 1e6)
csr_out = (torch.empty_like(rp), torch.empty_like(col), torch.empty_like(val))
ms = timeit(lambda: ops.coo_to_csr(n, n, *coo_out, rows_sorted=True, out=csr_out))
res["coo_to_csr_sorted"] = dict(ms=ms, alg_gbs=alg / ms / 1e6)
assert torch.equal(csr_out[0], rp) and torch.equal(csr_out[1], col)
ms = timeit(lambda: ops.degree_reorder(rp, True))
res["degree"] = dict(ms=ms, mrows_s=n / ms / 1e3, alg_gbs=(8 * n + 4) / ms / 1e6)
ms = timeit(lambda: ops.gray_row_keys(n, rp, col, 32, 10))
res["gray_keys"] = dict(ms=ms, mrows_s=n / ms / 1e3, alg_gbs=(4 * nnz + 8 * n) / ms / 1e6)
# shuffled COO sort
p = torch.randperm(nnz, device="cuda")
r0, c0, v0 = coo_out[0][p].contiguous(), coo_out[1][p].contiguous(), val[p].contiguous()
def sort_once():
    r, c, v = r0.clone(), c0.clone(), v0.clone()
    ops.coo_sort_(n, n, r, c, v)
ms_clone = timeit(lambda: (r0.clone(), c0.clone(), v0.clone()))
ms = timeit(sort_once) - ms_clone
res["coo_sort_shuffled"] = dict(ms=ms, mnnz_s=nnz / ms / 1e3)
# C2: 10M uniform random, sorted
row, colu, valu = synth.uniform_random_coo_torch(1 << 20, 1 << 20, args.coo_nnz, seed=3)
n2 = 1 << 20
o2 = (torch.empty(n2 + 1, dtype=torch.int32, device="cuda"), torch.empty_like(colu), torch.empty_like(valu))
ms = timeit(lambda: ops.coo_to_csr(n2, n2, row, colu, valu, rows_sorted=True, out=o2), reps=20)
alg = 20 * args.coo_nnz + 4 * (n2 + 1)
res["C2_coo_to_csr_10M"] = dict(ms=ms, alg_gbs=alg / ms / 1e6, frac_8tbs=alg / ms / 1e6 / 8000)
if "rcm" not in args.skip:
    try:
        ms = timeit(lambda: ops.rcm_reorder(rp, col), reps=3, warm=1)
        res["rcm"] = dict(ms=ms, mrows_s=n / ms / 1e3)
    except Exception as e:
        res["rcm"] = str(e)
print(json.dumps(res, indent=1))
