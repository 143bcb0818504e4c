#!/usr/bin/env python3
"""Phase ablation of k_permute_tile (SBX_DEBUG_TILE_STOP=k leaves the kernel after phase k; outputs are junk)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from sparsebase_amd import capi
    # SBX_DEBUG_TILE_STOP is live in the tuning build only (python -m sparsebase_amd.build --tuning)
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ.get('SBX_PROBE_LIB', 'tuning')}.so")
    from sparsebase_amd import ops, synth
    rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
    n, nnz = rp.numel() - 1, col.numel()
    val = torch.arange(nnz, device="cuda", dtype=torch.float32)
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.int32)
    out = (torch.empty_like(rp), torch.empty_like(col), torch.empty_like(val))
    for _ in range(2): ops.permute_csr(n, n, rp, col, val, perm, perm, out=out)
    torch.cuda.synchronize()
    ops.profile_enable(True)
    for _ in range(5): ops.permute_csr(n, n, rp, col, val, perm, perm, out=out)
    torch.cuda.synchronize()
    rep = ops.profile_report(); ops.profile_enable(False)
    print("RES", json.dumps({k: round(v[0] / 5, 3) for k, v in rep.items()}))
else:
    for stop in (1, 2, 3, 4, 5, 0):
        env = dict(os.environ, SBX_DEBUG_TILE_STOP=str(stop))
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RES")]
        print("stop", stop, line[0] if line else r.stdout[-500:] + r.stderr[-500:])
