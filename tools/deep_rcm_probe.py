#!/usr/bin/env python3
"""RCM on mesh-like inputs (grids, banded): GPU time next to the reference CPU path on the same box."""
import json, os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sparsebase_amd import ops, synth
import orc
ref = orc.Ref() if orc.ref_available() else None

def grid3d(a):
    idx = np.arange(a ** 3, dtype=np.int64).reshape(a, a, a)
    e = [np.stack([idx[:-1].ravel(), idx[1:].ravel()]), np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()]),
         np.stack([idx[:, :, :-1].ravel(), idx[:, :, 1:].ravel()])]
    e = np.concatenate(e, axis=1)
    src, dst = synth.symmetrize(e[0], e[1])
    return synth.csr_from_edges(a ** 3, src, dst)

cases = {"grid2d_2048": lambda: synth.grid_graph(2048, 2048),
         "grid2d_2048_shuffled": lambda: synth.grid_graph(2048, 2048, shuffle_seed=3),
         "grid3d_128": lambda: grid3d(128),
         "banded_1M_w64": lambda: tuple(t.cpu().numpy() for t in synth.banded_symmetric_torch(1 << 20, 64, per_row=12, seed=2))}
only = sys.argv[1:] or list(cases)
res = {}
for name in only:
    rp, col = cases[name]()
    n = len(rp) - 1
    drp, dcol = torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda()
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    ops.rcm_reorder(drp, dcol, out=out); torch.cuda.synchronize()
    t = time.perf_counter(); _, st = ops.rcm_reorder(drp, dcol, out=out, return_stats=True); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    r = dict(n=n, nnz=len(col), gpu_s=round(dt, 4), gpu_mrows_s=round(n / dt / 1e6, 2), levels=st["bfs_levels"], sweeps=st["bfs_sweeps"])
    if ref is not None:
        t = time.perf_counter(); want = ref.rcm_reorder(rp, col); cdt = time.perf_counter() - t
        r.update(cpu_s=round(cdt, 4), speedup=round(cdt / dt, 2), identical=bool(np.array_equal(out.cpu().numpy(), want)))
    res[name] = r
print(json.dumps(res, indent=1))
