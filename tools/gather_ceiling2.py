#!/usr/bin/env python3
"""Measured ceiling of the column relabel of Permute2D on this chip (replaces the figure borrowed from torch's gather
kernel): out[j] = table[col[j]] over the bench matrix's real col[] stream, tools/gather_replay.hip."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "libgather_replay.so")
src = os.path.join(ROOT, "tools", "gather_replay.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-shared", "-fPIC", "--offload-arch=gfx950", "-w", "-o", so, src])
if "--build-only" in sys.argv:
    sys.exit(0)
import torch
from sparsebase_amd import ops, synth
lib = C.CDLL(so)
lib.gather_replay.restype = C.c_float
lib.gather_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
n, nnz = rp.numel() - 1, col.numel()
rnd = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.int32)
rcm = ops.rcm_reorder(rp, col)
out = torch.empty_like(col)
# the permuted matrices' own column streams too (what a second permute, or the C4 shards, would gather through)
print(f"n {n} nnz {nnz}; table {4 * n / 1e6:.1f} MB")
best = {}
for tname, table in (("random order", rnd), ("rcm order", rcm)):
    for U in (4, 8, 16):
        for nt in (0, 1):
            for store in (1, 0):
                for wpc in (16, 32):
                    ms = lib.gather_replay(col.data_ptr(), table.data_ptr(), out.data_ptr(), nnz, U, nt, store, wpc, 5)
                    key = (tname, store)
                    if key not in best or ms < best[key][0]:
                        best[key] = (ms, U, nt, wpc)
                    print(f"{tname:13s} U={U:2d} nt={nt} store={store} waves/CU={wpc}: {ms:.3f} ms  {nnz / ms / 1e6:.0f} G gathers/s")
lib.gather_replay_full.restype = C.c_float
lib.gather_replay_full.argtypes = [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int]
val = torch.rand(nnz, device="cuda").view(torch.int32)
val_out = torch.empty_like(val)
for tname, table in (("random order", rnd), ("rcm order", rcm)):
    for wpc in (16, 32):
        ms = lib.gather_replay_full(col.data_ptr(), table.data_ptr(), out.data_ptr(), val.data_ptr(), val_out.data_ptr(), nnz, wpc, 5)
        print(f"FULL memory side of Permute2D ({tname}, waves/CU={wpc}): {ms:.3f} ms = {nnz / ms / 1e6:.0f} G entries/s = "
              f"{(16 * nnz + 12 * n) / (ms * 1e-3) / 8e12:.3f} of the 8 TB/s peak in Permute2D's algorithmic bytes")
# the same with the rows visited in a permuted order is what the real operation does; this replay streams them in order
for (tname, store), (ms, U, nt, wpc) in best.items():
    print(f"BEST {tname} store={store}: {ms:.3f} ms = {nnz / ms / 1e6:.0f} G/s (U={U} nt={nt} waves/CU={wpc})")
torch.cuda.synchronize()
