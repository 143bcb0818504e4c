#!/usr/bin/env python3
"""Every operation of the path on the GPU (data resident in HBM, median of warm runs) next to the REAL
reference on the host cores of the same box (one run, smaller instance where the full one takes minutes).
Prints JSON and a markdown table."""
import json, os, sys, time, statistics, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sparsebase_amd import capi
if os.environ.get("SBX_PROBE_LIB"):  # a variant built by tools/build_variant.py
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
from sparsebase_amd import ops, synth
import orc
ref = orc.Ref() if (orc.ref_available() and "--gpu-only" not in sys.argv) else None
HBM = 8000.0

def gpu_ms(f, reps=7):
    for _ in range(2): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)

def cpu_s(f):
    if ref is None: return None
    t = time.perf_counter(); f(); return time.perf_counter() - t

rows = []
def add(name, cfg, ms, alg_bytes, n_rows, cpu=None, cpu_cfg=None, cpu_rows=None):
    r = dict(op=name, config=cfg, gpu_ms=round(ms, 3), mrows_s=round(n_rows / ms / 1e3, 1))
    if alg_bytes:
        r["alg_gbs"] = round(alg_bytes / ms / 1e6, 1); r["frac_hbm"] = round(alg_bytes / ms / 1e6 / HBM, 4)
    if cpu is not None:
        r["cpu_ref_s"] = round(cpu, 3); r["cpu_config"] = cpu_cfg or cfg
        r["cpu_mrows_s"] = round((cpu_rows or n_rows) / cpu / 1e6, 2)
    rows.append(r); print(json.dumps(r), flush=True)

# ---------------- C2: 10 M-nnz uniform, n = m = 2^20
n = m = 1 << 20
row, col, val = synth.uniform_random_coo_torch(n, m, 10_000_000, seed=3)
nnz = col.numel()
hrow, hcol, hval = row.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
out = (torch.empty(n + 1, dtype=torch.int32, device="cuda"), torch.empty_like(col), torch.empty_like(val))
add("COO->CSR (copy)", "C2: 10 M nnz uniform", gpu_ms(lambda: ops.coo_to_csr(n, m, row, col, val, rows_sorted=True, out=out)),
    20 * nnz + 4 * (n + 1), n, cpu_s(lambda: ref.coo_to_csr(n, hrow, hcol, hval, m=m)))
rp, cc, vv = ops.coo_to_csr(n, m, row, col, val, rows_sorted=True)
hrp, hcc, hvv = rp.cpu().numpy(), cc.cpu().numpy(), vv.cpu().numpy()
out3 = (torch.empty_like(cc), torch.empty_like(cc), torch.empty_like(vv))
add("CSR->COO (copy)", "C2", gpu_ms(lambda: ops.csr_to_coo(n, m, rp, cc, vv, out=out3)), 20 * nnz + 4 * (n + 1), n,
    cpu_s(lambda: ref.csr_to_coo(hrp, hcc, hvv, m=m)))
perm = torch.randperm(nnz, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
srow, scol, sval = row[perm].contiguous(), col[perm].contiguous(), val[perm].contiguous()
def coo_sort():
    r, c, v = srow.clone(), scol.clone(), sval.clone()
    ops.coo_sort_(n, m, r, c, v)
t_clone = gpu_ms(lambda: (srow.clone(), scol.clone(), sval.clone()))
add("COO constructor sort (shuffled input)", "C2B", gpu_ms(coo_sort) - t_clone, 24 * nnz, n,
    cpu_s(lambda: ref.coo_sort(srow.cpu().numpy(), scol.cpu().numpy(), sval.cpu().numpy(), n=n, m=m)))
add("CSR->CSC", "C2", gpu_ms(lambda: ops.csr_to_csc(n, m, rp, cc, vv)), 20 * nnz + 4 * (n + 1) + 4 * (m + 1), n,
    cpu_s(lambda: ref.csr_to_csc(m, hrp, hcc, hvv)))
del row, col, val, srow, scol, sval, out, out3

# ---------------- C3: RMAT scale 22 (GPU) / scale 20 (CPU reference)
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
n, nnz = rp.numel() - 1, col.numel()
val = torch.arange(nnz, device="cuda", dtype=torch.float32)
crp_t, ccol_t = synth.rmat_symmetric_torch(20, 13, seed=1)
crp, ccol = crp_t.cpu().numpy(), ccol_t.cpu().numpy()
cn, cnnz = len(crp) - 1, len(ccol)
cval = np.arange(cnnz, dtype=np.float32)
ccfg = f"RMAT scale 20 ({cnnz / 1e6:.1f} M nnz)"
order = torch.empty(n, dtype=torch.int32, device="cuda")
t_rcm = cpu_s(lambda: ref.rcm_reorder(crp, ccol))
corder = ref.rcm_reorder(crp, ccol) if ref else None
add("RCMReorder", "C3: RMAT scale 22, 105 M nnz", gpu_ms(lambda: ops.rcm_reorder(rp, col, out=order), 5), None, n, t_rcm, ccfg, cn)
outp = (torch.empty_like(rp), torch.empty_like(col), torch.empty_like(val))
add("Permute2D (RCM order)", "C3", gpu_ms(lambda: ops.permute_csr(n, n, rp, col, val, order, order, out=outp)), 16 * nnz + 12 * n + 8, n,
    cpu_s(lambda: ref.permute_csr(crp, ccol, cval, corder, corder, m=cn)), ccfg, cn)
rnd = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.int32)
add("Permute2DRowWise (random order)", "C3", gpu_ms(lambda: ops.permute_csr(n, n, rp, col, val, rnd, None, out=outp)), 16 * nnz + 12 * n + 8, n,
    cpu_s(lambda: ref.permute_csr(crp, ccol, cval, synth.random_permutation(cn, 7), None, m=cn)), ccfg, cn)
add("DegreeReorder", "C3", gpu_ms(lambda: ops.degree_reorder(rp, True)), 8 * n + 4, n, cpu_s(lambda: ref.degree_reorder(crp, True, col=ccol)), ccfg, cn)
add("Bandwidth + Profile", "C3", gpu_ms(lambda: (ops.csr_bandwidth(rp, col), ops.csr_profile(rp, col))), 2 * (4 * nnz + 4 * n), n,
    cpu_s(lambda: ref.features(crp, ccol)), ccfg + ", all four features", cn)
add("GrayReorder device stage (keys)", "C3", gpu_ms(lambda: ops.gray_row_keys(n, rp, col, 32, 10)), 4 * nnz + 16 * n, n)
add("GrayReorder, ordering on the device (opt-in, stable ties)", "C3", gpu_ms(lambda: ops.gray_reorder(n, rp, col, 32, 10, 4)),
    4 * nnz + 16 * n, n)
# the COO constructor's sort at C3 size: the bench matrix's entries, shuffled
rows_c3 = torch.repeat_interleave(torch.arange(n, device="cuda", dtype=torch.int32), (rp[1:] - rp[:-1]).long())
p3 = torch.randperm(nnz, device="cuda", generator=torch.Generator(device="cuda").manual_seed(11))
s3 = (rows_c3[p3].contiguous(), col[p3].contiguous(), val[p3].contiguous())
del rows_c3, p3
w3 = tuple(torch.empty_like(t) for t in s3)
def coo_sort_c3():
    for d, t in zip(w3, s3): d.copy_(t)
    ops.coo_sort_(n, n, *w3)
def clone_c3():
    for d, t in zip(w3, s3): d.copy_(t)
add("COO constructor sort (shuffled input)", "C3: 105 M entries", gpu_ms(coo_sort_c3, 5) - gpu_ms(clone_c3, 5), 24 * nnz, n)
del outp, val, s3, w3

# ---------------- C5: banded, Gray device stage (+ CPU reference of the whole GrayReorder on a 1 M-row instance)
for w, tag in ((64, "C5: banded +-64, n = 4 M"), ((1 << 22) // 16, "C5: banded +-m/16")):
    brp, bcol = synth.banded_symmetric_torch(1 << 22, w, per_row=12, seed=2)
    bn, bnnz = brp.numel() - 1, bcol.numel()
    t = None
    if ref is not None and w == 64:
        srp, scol = (x.cpu().numpy() for x in synth.banded_symmetric_torch(1 << 20, 64, per_row=12, seed=2))
        t = cpu_s(lambda: ref.gray_reorder(srp, scol, 1 << 20, 32, 10, 4))
    add("GrayReorder device stage (keys)", tag, gpu_ms(lambda: ops.gray_row_keys(bn, brp, bcol, 32, 10)), 4 * bnnz + 8 * bn + 4, bn,
        t, "whole GrayReorder, banded +-64, n = 1 M", 1 << 20)
    del brp, bcol

print(json.dumps(rows))
print("\n| operation | config | GPU time | Mrows/s | algorithmic GB/s | of 8 TB/s | reference on the host (config) | host Mrows/s |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r['op']} | {r['config']} | {r['gpu_ms']} ms | {r['mrows_s']} | {r.get('alg_gbs', '—')} | "
          f"{('%.1f %%' % (100 * r['frac_hbm'])) if 'frac_hbm' in r else '—'} | "
          f"{(str(r['cpu_ref_s']) + ' s (' + r['cpu_config'] + ')') if 'cpu_ref_s' in r else '—'} | {r.get('cpu_mrows_s', '—')} |")
