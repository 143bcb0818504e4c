#!/usr/bin/env python3
"""Permute2D on the bench matrix, random and RCM order: whole-call time, per-kernel-group times (library profiler) and a
digest of the result (equal digests = bit-identical outputs across builds).  SBX_PROBE_LIB=<name> picks a variant
library (tools/build_variant.py)."""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparsebase_amd import capi
if os.environ.get("SBX_PROBE_LIB"):
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
from sparsebase_amd import ops, synth

def digest(ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.cpu().numpy().tobytes())
    return h.hexdigest()[:16]

scale = int(os.environ.get("PROBE_SCALE", "22"))
rp, col = synth.rmat_symmetric_torch(scale, 13, seed=1)
n, nnz = rp.numel() - 1, col.numel()
val = torch.rand(nnz, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
rnd = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.int32)
rcm = ops.rcm_reorder(rp, col)
tag = os.environ.get("SBX_PROBE_LIB", "product")
res = {"tag": tag}
for name, perm in (("random", rnd), ("rcm", rcm)):
    out = (torch.empty_like(rp), torch.empty_like(col), torch.empty_like(val))
    for _ in range(3):
        ops.permute_csr(n, n, rp, col, val, perm, perm, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        ops.permute_csr(n, n, rp, col, val, perm, perm, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 20 * 1e3
    ops.profile_enable(True)
    for _ in range(5):
        ops.permute_csr(n, n, rp, col, val, perm, perm, out=out)
    torch.cuda.synchronize()
    prof = ops.profile_report()
    ops.profile_enable(False)
    groups = {k: round(v[0] / 5, 4) for k, v in prof.items() if v[0] > 0}
    res[name] = {"ms": round(ms, 4), "digest": digest(out), "groups": groups}
    print(f"{tag} {name}: {ms:.3f} ms  frac_of_hbm_peak {(16 * nnz + 12 * n) / (ms * 1e-3) / 8e12:.3f}  digest {res[name]['digest']}  {groups}", flush=True)
if "--pattern" in sys.argv:
    for name, v in (("pattern", None), ("f64", torch.rand(nnz, device="cuda", dtype=torch.float64))):
        out = (torch.empty_like(rp), torch.empty_like(col), None if v is None else torch.empty_like(v))
        for _ in range(3):
            ops.permute_csr(n, n, rp, col, v, rnd, rnd, out=out)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            ops.permute_csr(n, n, rp, col, v, rnd, rnd, out=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 10 * 1e3
        res[name] = {"ms": round(ms, 4), "digest": digest([o for o in out if o is not None])}
        print(f"{tag} {name} (random order): {ms:.3f} ms digest {res[name]['digest']}", flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "v3_probe.jsonl"), "a") as f:
    f.write(json.dumps(res) + "\n")
