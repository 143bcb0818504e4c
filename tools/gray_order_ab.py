import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from sparsebase_amd import capi
if os.environ.get("SBX_PROBE_LIB"):  # a variant built by tools/build_variant.py
    capi.LIB_PATH = os.path.join("/root/repo", "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
from sparsebase_amd import ops, synth
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
n = rp.numel() - 1
def t(f, k=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
print(os.environ.get("SBX_GRAY_ORDER_THREE_SORTS", "0"), "keys %.3f ms  reorder %.3f ms" % (t(lambda: ops.gray_row_keys(n, rp, col, 32, 10)), t(lambda: ops.gray_reorder(n, rp, col, 32, 10, 4))))
ops.profile_enable(True)
for _ in range(5): ops.gray_reorder(n, rp, col, 32, 10, 4)
torch.cuda.synchronize()
rep = ops.profile_report(); ops.profile_enable(False)
print(" ".join(f"{k} {v[0] / 5:.3f}" for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0])[:8]))
