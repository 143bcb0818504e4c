#!/usr/bin/env python3
"""tools/replay_policy.hip at the size of one C4 shard: 146 M entries relabelled through a 33.5 M-entry table (134 MB:
beyond the L2s, inside the 256 MB of infinity cache) — streaming loads x streaming stores under plain / nt / sc1 /
sc0 sc1 / sc0 sc1 nt.  Does any policy keep the table in the infinity cache while 2.3 GB stream past?"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "libreplay_policy.so")
src = os.path.join(ROOT, "tools", "replay_policy.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-shared", "-fPIC", "--offload-arch=gfx950", "-w", "-o", so, src])
import torch
lib = C.CDLL(so)
lib.replay_policy.restype = C.c_float
lib.replay_policy.argtypes = [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
m, nnz = 1 << 25, 146_000_000
g = torch.Generator(device="cuda").manual_seed(5)
table = torch.randperm(m, device="cuda", generator=g).to(torch.int32)
col = torch.randint(0, m, (nnz,), device="cuda", generator=g, dtype=torch.int32)
val = torch.rand(nnz, device="cuda").view(torch.int32)
out, val_out = torch.empty_like(col), torch.empty_like(val)
names = ["plain", "nt", "sc1", "sc0 sc1", "sc0 sc1 nt"]
print(f"table {m} entries ({m * 4 >> 20} MB), {nnz} entries; loads x stores, ms")
print("loads \\ stores".ljust(14) + "".join(s.rjust(12) for s in names))
for lp in range(5):
    row = names[lp].ljust(14)
    for sp in range(5):
        ms = lib.replay_policy(col.data_ptr(), table.data_ptr(), out.data_ptr(), val.data_ptr(), val_out.data_ptr(), nnz, lp, sp,
                               16, 3)
        row += f"{ms:12.3f}"
    print(row, flush=True)
torch.cuda.synchronize()
assert torch.equal(out, table[col.long()])
