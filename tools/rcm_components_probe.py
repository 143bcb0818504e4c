#!/usr/bin/env python3
"""RCM on inputs dominated by many mid-size components / deep narrow bands: GPU time next to the real reference
(or the restatement) on the host (diagnostic; verdict r1 #6)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import orc
from sparsebase_amd import ops, synth
from test_gpu_parity import _many_components
impl = orc.Ref() if orc.ref_available() else orc.Oracle()
out = {}
cases = {"20k components of 65..420": lambda: _many_components(20000, 5),
         "100k components of 65..200": lambda: _many_components(100000, 8, 65, 200),
         "random band +-64, 1M rows": lambda: synth.banded_symmetric(1 << 20, 64, per_row=12, seed=2)}
for name, make in cases.items():
    rp, col = make()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    drp, dcol = d(rp), d(col)
    got = ops.rcm_reorder(drp, dcol); torch.cuda.synchronize()
    t = time.perf_counter(); got = ops.rcm_reorder(drp, dcol); torch.cuda.synchronize(); gpu = time.perf_counter() - t
    t = time.perf_counter(); want = impl.rcm_reorder(rp, col); cpu = time.perf_counter() - t
    out[name] = dict(n=len(rp) - 1, nnz=len(col), gpu_s=round(gpu, 4), host_s=round(cpu, 4), speedup=round(cpu / gpu, 2),
                     identical=bool(np.array_equal(got.cpu().numpy(), want)), host=type(impl).__name__)
print(json.dumps(out, indent=1))
