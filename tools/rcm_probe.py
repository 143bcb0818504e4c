#!/usr/bin/env python3
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebase_amd import ops, synth
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
out = torch.empty(rp.numel() - 1, dtype=torch.int32, device="cuda")
def run():
    try: ops.rcm_reorder(rp, col, out=out)
    except Exception as e: pass
run(); torch.cuda.synchronize()
ops.profile_enable(True)
for _ in range(3): run()
rep = ops.profile_report()
print(os.environ.get("SBX_DEBUG_BU_MODE", "0"), json.dumps({k: (round(v[0] / 3, 3), v[1] // 3) for k, v in rep.items() if k.startswith("bfs")}))
