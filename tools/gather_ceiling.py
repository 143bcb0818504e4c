#!/usr/bin/env python3
"""What a pure relabel (out[j] = table[col[j]]) of the bench matrix's column array costs with torch's gather: the
ceiling any Permute2D with a column map lives under (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparsebase_amd import synth
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
n, nnz = rp.numel() - 1, col.numel()
perm = torch.randperm(n, device="cuda").to(torch.int32)
uni = torch.randint(0, n, (nnz,), device="cuda", dtype=torch.int32)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for name, idx in (("rmat columns", col), ("uniform", uni), ("rmat columns, shuffled", col[torch.randperm(nnz, device="cuda")])):
    ms = t(lambda: torch.index_select(perm, 0, idx))
    print(f"{name}: {ms:.3f} ms  {nnz / ms / 1e6:.1f} G gathers/s")
small = torch.arange(1 << 16, device="cuda", dtype=torch.int32)
ms = t(lambda: torch.index_select(small, 0, uni & 0xFFFF))
print(f"uniform into 256 KB table: {ms:.3f} ms {nnz / ms / 1e6:.1f} G/s (includes the mask kernel)")
ms = t(lambda: perm[: nnz % n + n // 2].clone())
