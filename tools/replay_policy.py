#!/usr/bin/env python3
"""Cache-policy sweep of Permute2D's memory side on the bench matrix (tools/replay_policy.hip): streaming loads x
streaming stores under plain / nt / sc1 / sc0 sc1 / sc0 sc1 nt, results checked."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "libreplay_policy.so")
src = os.path.join(ROOT, "tools", "replay_policy.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-shared", "-fPIC", "--offload-arch=gfx950", "-w", "-o", so, src])
if "--build-only" in sys.argv:
    sys.exit(0)
import torch
from sparsebase_amd import ops, synth
lib = C.CDLL(so)
lib.replay_policy.restype = C.c_float
lib.replay_policy.argtypes = [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
n, nnz = rp.numel() - 1, col.numel()
table = ops.rcm_reorder(rp, col)
val = torch.rand(nnz, device="cuda").view(torch.int32)
out = torch.empty_like(col)
val_out = torch.empty_like(val)
want = table[col.long()]
names = ["plain", "nt", "sc1", "sc0 sc1", "sc0 sc1 nt"]
print(f"n {n} nnz {nnz}; loads x stores, ms (rcm order table, 16 waves/CU)")
print("loads \\ stores".ljust(14) + "".join(s.rjust(12) for s in names))
for lp in range(5):
    row = names[lp].ljust(14)
    for sp in range(5):
        out.zero_()
        ms = lib.replay_policy(col.data_ptr(), table.data_ptr(), out.data_ptr(), val.data_ptr(), val_out.data_ptr(), nnz, lp, sp, 16, 5)
        torch.cuda.synchronize()
        ok = bool((out == want).all()) and bool((val_out == val).all())
        row += f"{ms:10.3f}{'' if ok else '!'}  "
    print(row, flush=True)
