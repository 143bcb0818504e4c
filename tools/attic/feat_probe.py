import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from sparsebase_amd import ops, synth
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
torch.cuda.synchronize()
for name, f in (("bandwidth", lambda: ops.csr_bandwidth(rp, col)), ("profile", lambda: ops.csr_profile(rp, col))):
    for _ in range(3): f()
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    print(name, "wall ms", sorted(ts)[3] * 1e3)
ops.profile_enable(True)
for _ in range(5):
    ops.csr_bandwidth(rp, col); ops.csr_profile(rp, col)
print(ops.profile_report())
