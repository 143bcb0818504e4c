#!/usr/bin/env python3
"""Config C4 on one GPU: one row-range shard (1/8 of the new rows) of Permute2D on a ~1 B-nnz RMAT CSR.
Reports the shard time — what each of the 8 ranks does between the two all-gathers."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebase_amd import ops, synth, sharded
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 25
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 18
t = time.perf_counter()
rp, col = synth.rmat_symmetric_torch(scale, ef, seed=1)
torch.cuda.synchronize()
gen_s = time.perf_counter() - t
n, nnz = rp.numel() - 1, col.numel()
val = torch.ones(nnz, device="cuda", dtype=torch.float32)
perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.int32)
world = 8
# the scratch arena grows (and is consolidated) inside the calls that first need more: reserve it up front, as a caller
# that knows its sizes would (sbx_reserve), so that no timed shard pays for hipMalloc
ops.handle_for(torch.device("cuda", torch.cuda.current_device())).reserve(int(os.environ.get("C4_RESERVE_GB", "6")) << 30)
res = dict(n=n, nnz=nnz, generate_s=round(gen_s, 1), shards={})
for name, ranges in (("rows/8", sharded.row_ranges(n, world)),):
    times = []
    for r in (0, 3, 7):
        a, b = ranges[r]
        ops.permute_csr_rows(n, n, rp, col, val, perm, perm, a, b, capacity=nnz // 4)
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = ops.permute_csr_rows(n, n, rp, col, val, perm, perm, a, b, capacity=nnz // 4)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        k = out[1].numel()
        times.append(dict(rank=r, rows=b - a, shard_nnz=k, ms=round(dt * 1e3, 3),
                          alg_gbs=round((16 * k + 12 * (b - a)) / dt / 1e9, 1)))
        del out
    res["shards"][name] = times
# row-wise variant of the same shard
a, b = sharded.row_ranges(n, world)[3]
ops.permute_csr_rows(n, n, rp, col, val, perm, None, a, b, capacity=nnz // 4); torch.cuda.synchronize()
t = time.perf_counter(); out = ops.permute_csr_rows(n, n, rp, col, val, perm, None, a, b, capacity=nnz // 4); torch.cuda.synchronize()
dt = time.perf_counter() - t
res["rowwise_shard3"] = dict(shard_nnz=out[1].numel(), ms=round(dt * 1e3, 3), alg_gbs=round(16 * out[1].numel() / dt / 1e9, 1))
print(json.dumps(res, indent=1))
# where a shard's time goes: every launch bracketed by HIP events, the paths back to back (SBX_PERMUTE_OVERLAP=0 style)
if os.environ.get("C4_PROFILE"):
    a, b = sharded.row_ranges(n, world)[3]
    ops.profile_enable(True)
    for _ in range(3):
        ops.permute_csr_rows(n, n, rp, col, val, perm, perm, a, b, capacity=nnz // 4)
    torch.cuda.synchronize()
    rep = ops.profile_report()
    ops.profile_enable(False)
    print(json.dumps({k: [round(v[0] / 3, 3), v[1] // 3] for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0])}))
