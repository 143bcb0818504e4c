#!/usr/bin/env python3
"""Relabel gather under different cache policies of the table load (tools/gather_policy.hip), real col[] stream."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "libgather_policy.so")
src = os.path.join(ROOT, "tools", "gather_policy.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-shared", "-fPIC", "--offload-arch=gfx950", "-w", "-o", so, src])
if "--build-only" in sys.argv:
    sys.exit(0)
import torch
from sparsebase_amd import synth
lib = C.CDLL(so)
lib.gather_policy.restype = C.c_float
lib.gather_policy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int]
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
n, nnz = rp.numel() - 1, col.numel()
table = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.int32)
uni = torch.randint(0, n, (nnz,), device="cuda", dtype=torch.int32)
out = torch.empty_like(col)
names = ["plain", "nt", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc0", "ushort"]
want = table[col.long()]
for sname, idx in (("rmat col stream", col), ("uniform indices", uni)):
    for p, nm in enumerate(names):
        for wpc in (16, 32):
            ms = lib.gather_policy(idx.data_ptr(), table.data_ptr(), out.data_ptr(), nnz, p, wpc, 5)
            ok = bool(torch.equal(out, want)) if (idx is col and p != 6) else None
            print(f"{sname:16s} {nm:11s} waves/CU={wpc}: {ms:.3f} ms {nnz / ms / 1e6:.0f} G/s ok={ok}", flush=True)
