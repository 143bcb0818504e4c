#!/usr/bin/env python3
"""64-bit index arrays (the reference's <int64,int64,double> tuple) through the native kernels: Permute2D on the bench
matrix (random and RCM order) and the COO constructor's sort of C2B, timed next to the 32-bit calls.  Under
`rocprofv3 --kernel-trace --stats` (tools/int64_trace.sh) the kernel list shows no k_narrow / k_widen."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparsebase_amd import ops, synth

def timed(f, reps):
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3

reps = int(os.environ.get("REPS", "10"))
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
n, nnz = rp.numel() - 1, col.numel()
val = torch.rand(nnz, device="cuda", dtype=torch.float64)
orders = {"random": torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.int32),
          "rcm": ops.rcm_reorder(rp, col)}
for it in (torch.int32, torch.int64):
    a, b = rp.to(it), col.to(it)
    out = (torch.empty_like(a), torch.empty_like(b), torch.empty_like(val))
    for name, perm in orders.items():
        p = perm.to(it)
        ms = timed(lambda: ops.permute_csr(n, n, a, b, val, p, p, out=out), reps)
        print("permute2d f64 values, %s indices, %s order: %.3f ms" % (str(it)[6:], name, ms), flush=True)
for it in (torch.int32, torch.int64):  # RCM reads and writes 64-bit arrays itself (sbx_rcm64.hip): no narrowed copies
    a, b = rp.to(it), col.to(it)
    o = torch.empty(n, dtype=it, device="cuda")
    print("rcm, %s indices: %.3f ms" % (str(it)[6:], timed(lambda: ops.rcm_reorder(a, b, out=o), reps)), flush=True)
for it in (torch.int32, torch.int64):  # the Gray key stage likewise (sbx_gray64.hip)
    a, b = rp.to(it), col.to(it)
    print("gray keys, %s indices: %.3f ms" % (str(it)[6:], timed(lambda: ops.gray_row_keys(n, a, b, 32, 10), reps)), flush=True)
row, c2, v2 = synth.uniform_random_coo_torch(1 << 20, 1 << 20, 10_000_000, seed=3)
sh = torch.randperm(row.numel(), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
for it in (torch.int32, torch.int64):
    r0, c0, v0 = row[sh].to(it).contiguous(), c2[sh].to(it).contiguous(), v2[sh].contiguous()
    r, c, v = r0.clone(), c0.clone(), v0.clone()
    def once():
        r.copy_(r0); c.copy_(c0); v.copy_(v0)
        ops.coo_sort_(1 << 20, 1 << 20, r, c, v)
    def copies():
        r.copy_(r0); c.copy_(c0); v.copy_(v0)
    ms = timed(once, reps) - timed(copies, reps)
    print("coo sort C2B, %s indices: %.3f ms" % (str(it)[6:], ms), flush=True)
