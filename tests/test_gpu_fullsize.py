"""Full-size BASELINE configurations on the GPU (C2, C3, C5): bit-exact comparison with the oracle
where the oracle finishes in seconds, otherwise size-independent properties (permutation-ness,
round trips, sortedness, idempotence, checksums)."""
import os

import numpy as np
import pytest

from sparsebase_amd import synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no GPU is visible")
    from sparsebase_amd import ops as _ops
    return _ops


def test_c2_coo_to_csr_10m_uniform(ops, oracle):
    n = m = 1 << 20
    row, col, val = synth.uniform_random_coo_torch(n, m, 10_000_000, seed=3)
    rp, co, vo = ops.coo_to_csr(n, m, row, col, val, rows_sorted=True)
    want = oracle.coo_to_csr(n, row.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy())
    assert np.array_equal(rp.cpu().numpy(), want[0])
    assert np.array_equal(co.cpu().numpy(), want[1]) and np.array_equal(vo.cpu().numpy(), want[2])
    # without the hint (sortedness detected on the device) and as a move conversion
    assert torch.equal(ops.coo_to_csr(n, m, row, col, val)[0], rp)
    assert torch.equal(ops.coo_to_csr(n, m, row, col, None, move=True)[0], rp)
    # CSR -> COO round trip, and the COO-constructor sort of a shuffled copy (config 2B)
    back = ops.csr_to_coo(n, m, rp, co, vo)
    assert torch.equal(back[0], row) and torch.equal(back[1], col) and torch.equal(back[2], val)
    p = torch.randperm(row.numel(), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    r2, c2, v2 = row[p].contiguous(), col[p].contiguous(), val[p].contiguous()
    assert not ops.coo_is_sorted(r2, c2)
    ops.coo_sort_(n, m, r2, c2, v2)
    assert torch.equal(r2, row) and torch.equal(c2, col) and torch.equal(v2, val)


@pytest.fixture(scope="module")
def c3():
    rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
    val = (torch.arange(col.numel(), device="cuda", dtype=torch.int32) % 1021).to(torch.float32)
    return rp, col, val


def test_c3_rcm_100m_bit_exact_vs_oracle(ops, oracle, c3):
    rp, col, _ = c3
    order, stats = ops.rcm_reorder(rp, col, return_stats=True)
    want = oracle.rcm_reorder(rp.cpu().numpy(), col.cpu().numpy())
    assert np.array_equal(order.cpu().numpy(), want), stats


def test_c3_permute_100m_properties(ops, c3):
    rp, col, val = c3
    n, nnz = rp.numel() - 1, col.numel()
    order = ops.rcm_reorder(rp, col)
    assert torch.equal(torch.sort(order.long()).values, torch.arange(n, device="cuda"))  # a permutation
    prp, pcol, pval = ops.permute_csr(n, n, rp, col, val, order, order)
    assert int(prp[-1]) == nnz and int(prp[0]) == 0
    assert ops.csr_rows_sorted(prp, pcol)                                   # every row sorted
    deg_old = (rp[1:] - rp[:-1]).long()
    deg_new = (prp[1:] - prp[:-1]).long()
    assert torch.equal(deg_new[order.long()], deg_old)                     # row lengths moved with the rows
    assert int(pcol.long().sum()) == int(order.long()[col.long()].sum())   # column relabelling checksum
    assert float(pval.double().sum()) == float(val.double().sum())         # payload multiset checksum
    before = pcol.clone()
    ops.csr_sort_rows_(n, n, prp, pcol, pval)                               # idempotent
    assert torch.equal(before, pcol)
    inv = ops.inverse_permutation(order)
    brp, bcol, bval = ops.permute_csr(n, n, prp, pcol, pval, inv, inv)     # inverse restores the matrix
    assert torch.equal(brp, rp) and torch.equal(bcol, col) and torch.equal(bval, val)
    ro, co, vo = ops.csr_to_coo(n, n, prp, pcol, pval)                     # CSR -> COO -> CSR round trip
    assert ops.coo_is_sorted(ro, co)
    rrp, rcol, rval = ops.coo_to_csr(n, n, ro, co, vo, rows_sorted=True)
    assert torch.equal(rrp, prp) and torch.equal(rcol, pcol) and torch.equal(rval, pval)
    # degree reorder: non-decreasing degrees along the new order
    dinv = ops.degree_reorder(rp, True)
    perm = torch.empty_like(dinv)
    perm[dinv.long()] = torch.arange(n, device="cuda", dtype=torch.int32)
    assert bool((deg_old[perm.long()][1:] >= deg_old[perm.long()][:-1]).all())


def test_c3_permute_100m_bit_exact_vs_oracle(ops, oracle, c3):
    """The headline operation inside the suite, bit for bit: Permute2D(order, order) of the bench matrix
    (permute/permute_order_two.cc:23-79 + the constructor's row sort, format/csr.cc:118-157) for the RCM order the bench
    step uses and for one random order, float32 values, against the oracle's restatement on the host."""
    rp, col, val = c3
    n = rp.numel() - 1
    hrp, hcol, hval = rp.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
    rnd = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.int32)
    for name, order in (("rcm", ops.rcm_reorder(rp, col)), ("random", rnd)):
        prp, pcol, pval = ops.permute_csr(n, n, rp, col, val, order, order)
        horder = order.cpu().numpy()
        wrp, wcol, wval = oracle.permute_csr(hrp, hcol, hval, horder, horder)
        assert np.array_equal(prp.cpu().numpy(), wrp), name
        assert np.array_equal(pcol.cpu().numpy(), wcol), name
        assert np.array_equal(pval.cpu().numpy().view(np.uint32), wval.view(np.uint32)), name
        del prp, pcol, pval, wrp, wcol, wval


def test_c3_csc_and_features_100m(ops, oracle, c3):
    """§8(f) rows at full size: transpose twice = identity, the transpose of a symmetric pattern has the same
    row_ptr/col, features against the oracle, and RCM lowers the profile of the power-law matrix."""
    rp, col, val = c3
    n, nnz = rp.numel() - 1, col.numel()
    cp, ro, vo = ops.csr_to_csc(n, n, rp, col, val)
    assert torch.equal(cp, rp) and torch.equal(ro, col)                    # structurally symmetric input
    assert float(vo.double().sum()) == float(val.double().sum())
    back = ops.csr_to_csc(n, n, cp, ro, vo)
    assert torch.equal(back[0], rp) and torch.equal(back[1], col) and torch.equal(back[2], val)
    row = ops.csr_to_coo(n, n, rp, col, None, move=True)[0]
    cp2, ro2, vo2 = ops.coo_to_csc(n, n, row, col, val)
    assert torch.equal(cp2, cp) and torch.equal(ro2, ro) and torch.equal(vo2, vo)
    hrp, hcol = rp.cpu().numpy(), col.cpu().numpy()
    assert ops.csr_bandwidth(rp, col) == oracle.csr_bandwidth(hrp, hcol)
    before = ops.csr_profile(rp, col)
    assert before == oracle.csr_profile(hrp, hcol)
    deg = ops.csr_degrees(rp)
    assert int(deg.long().sum()) == nnz
    dist = ops.csr_degree_distribution(rp, nnz, torch.float64)
    assert abs(float(dist.sum()) - 1.0) < 1e-9
    order = ops.rcm_reorder(rp, col)
    prp, pcol, _ = ops.permute_csr(n, n, rp, col, None, order, order)
    assert ops.csr_profile(prp, pcol) < before


def test_c3_gray_keys_degree_and_coo_sort_100m_bit_exact(ops, oracle, c3):
    """The power-law kernels of round 3 at full size, bit for bit against the oracle: the Gray key stage on the bench
    matrix (the banded kernel stops, k_gray_rows_balanced / k_gray_rows_medium / k_gray_units_finish take over; hubs of
    hundreds of 1024-entry units) at three parameter sets, DegreeReorder in both directions (one digit pass + the
    re-sorted tail of 76 K vertices), and the hybrid COO constructor sort on the matrix's 105 M entries shuffled
    (three digit passes, groups = rows, the hubs on the long-group path)."""
    rp, col, val = c3
    n, nnz = rp.numel() - 1, col.numel()
    hrp, hcol = rp.cpu().numpy(), col.cpu().numpy()
    for res, thr in ((32, 10), (16, 20), (64, 2)):
        deg, key, counts = ops.gray_row_keys(n, rp, col, res, thr)
        wdeg, wkey, wcounts = oracle.gray_row_keys(hrp, hcol, n, res, thr)
        assert np.array_equal(deg.cpu().numpy(), wdeg), (res, thr)
        assert np.array_equal(key.cpu().numpy().view(np.uint64), wkey), (res, thr)
        assert list(counts) == wcounts.tolist(), (res, thr)
    for asc in (True, False):
        assert np.array_equal(ops.degree_reorder(rp, asc).cpu().numpy(), oracle.degree_reorder(hrp, asc))
    row = ops.csr_to_coo(n, n, rp, col, None, move=True)[0]
    p = torch.randperm(nnz, device="cuda", generator=torch.Generator(device="cuda").manual_seed(11))
    r, c, v = row[p].contiguous(), col[p].contiguous(), val[p].contiguous()
    ops.coo_sort_(n, n, r, c, v)          # distinct coordinates: the sorted COO is the CSR's own expansion
    assert torch.equal(r, row) and torch.equal(c, col) and torch.equal(v, val)


def test_c2b_coo_sort_10m_with_duplicates(ops, oracle):
    """C2B: 10 M uniform entries over 2^20 x 2^20, shuffled, a tenth of them duplicated coordinates with different
    values: the hybrid sort (two digit passes, 65 536 groups of ~150 records sorted in LDS) keeps duplicates in input
    order like the oracle's stable sort."""
    n = m = 1 << 20
    row, col, val = synth.uniform_random_coo_torch(n, m, 10_000_000, seed=3, shuffled=True)
    k = row.numel() // 10
    row[:k], col[:k] = row[k:2 * k].clone(), col[k:2 * k].clone()
    hr, hc, hv = row.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
    r, c, v = row.clone(), col.clone(), val.clone()
    ops.coo_sort_(n, m, r, c, v)
    wr, wc, wv = oracle.coo_sort(hr, hc, hv)
    assert np.array_equal(r.cpu().numpy(), wr) and np.array_equal(c.cpu().numpy(), wc) and np.array_equal(v.cpu().numpy(), wv)


@pytest.mark.parametrize("half_bandwidth", [64, (1 << 22) // 16])
def test_c5_gray_keys_100m_banded(ops, oracle, half_bandwidth):
    n = 1 << 22
    rp, col = synth.banded_symmetric_torch(n, half_bandwidth, per_row=12, seed=2)
    for res, thr in ((32, 10), (16, 20)):
        deg, key, counts = ops.gray_row_keys(n, rp, col, res, thr)
        wdeg, wkey, wcounts = oracle.gray_row_keys(rp.cpu().numpy(), col.cpu().numpy(), n, res, thr)
        assert np.array_equal(deg.cpu().numpy(), wdeg)
        assert np.array_equal(key.cpu().numpy().view(np.uint64), wkey)
        assert list(counts) == wcounts.tolist()


@pytest.mark.parametrize("half_bandwidth", [64, (1 << 22) // 16])
def test_c5_gray_end_to_end_100m_banded(oracle, tmp_path, half_bandwidth):
    """BASELINE config 5 END TO END: GrayReorder through the C++ API (device key stage + the host ordering stage over
    4 M row keys, host/bin/reorder_cli) against the oracle's whole ordering, both half-bandwidths of SURVEY §8d
    (w = 64 takes the reference's "highly banded" early-out, w = m/16 the full bitmap path) and both parameter sets
    ((BitSize32, 10, 4) and the example's (BitSize16, 20, (N/n)/16), examples/gray_order/gray_order.cc:52-54)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "sparsebase_amd", "host", "bin", "reorder_cli")
    if not os.path.exists(cli):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "sparsebase_amd", "host"), "all"])
    n = 1 << 22
    rp, col = (t.cpu().numpy() for t in synth.banded_symmetric_torch(n, half_bandwidth, per_row=12, seed=2))
    a, b, o = (str(tmp_path / x) for x in ("rp.bin", "col.bin", "out.bin"))
    rp.tofile(a)
    col.tofile(b)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(root, "sparsebase_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    for params in ((32, 10, 4), (16, 20, max(1, (len(col) // n) // 16))):
        subprocess.run([cli, "gray", a, b, o, str(n), str(n), *map(str, params), "--device"], check=True, env=env,
                       timeout=900)
        got = np.fromfile(o, np.int32)
        want = oracle.gray_reorder(rp, col, n, *params)
        assert np.array_equal(got, want), (half_bandwidth, params)


def _check_gray_device_ordering(ops, oracle, rp, col, n, params, idx64):
    """sbx_gray_reorder (stable ties) against the numpy statement of that order row for row, and against the
    reference-exact oracle on every row whose place the reference's comparators decide."""
    from test_gpu_parity import _gray_stable_model
    res, thr, grp = params
    hrp, hcol = rp.cpu().numpy(), col.cpu().numpy()
    drp, dcol = (rp.long(), col.long()) if idx64 else (rp, col)
    inv_t = ops.gray_reorder(n, drp, dcol, res, thr, grp)
    assert inv_t.dtype == (torch.int64 if idx64 else torch.int32)
    inv = inv_t.cpu().numpy().astype(np.int64)
    assert np.array_equal(np.sort(inv), np.arange(n))
    deg, key, counts = oracle.gray_row_keys(hrp, hcol, n, res, thr)
    model, group = _gray_stable_model(deg, key, counts, min(res, n), thr, grp)
    assert np.array_equal(inv, model), (params, idx64)
    # rows with a unique (class, section, key): the reference's comparators decide their place
    g = np.ascontiguousarray(group).view([("", group.dtype)] * group.shape[1]).ravel()
    _, first, cnt = np.unique(g, return_index=True, return_counts=True)
    decided = first[cnt == 1]
    want = oracle.gray_reorder(hrp, hcol, n, res, thr, grp).astype(np.int64)
    assert np.array_equal(inv[decided], want[decided]), (params, idx64, len(decided))
    return len(decided)


@pytest.mark.parametrize("idx64", [False, True])
def test_c3_gray_device_ordering_100m(ops, oracle, c3, idx64):
    """sbx_gray_reorder at the size the bench line times it (4.2 M rows, 105 M nnz: the radix sorts take their one-sweep
    paths, the key stage its power-law kernels), 32- and 64-bit index arrays."""
    rp, col, _ = c3
    n = rp.numel() - 1
    decided = _check_gray_device_ordering(ops, oracle, rp, col, n, (32, 10, 4), idx64)
    assert decided > 1000


@pytest.mark.parametrize("half_bandwidth,idx64", [(64, False), ((1 << 22) // 16, False), ((1 << 22) // 16, True)])
def test_c5_gray_device_ordering_100m_banded(ops, oracle, half_bandwidth, idx64):
    """The same at BASELINE config 5 (both half-bandwidths: the "highly banded" early-out and the full bitmap path)."""
    n = 1 << 22
    rp, col = synth.banded_symmetric_torch(n, half_bandwidth, per_row=12, seed=2)
    for params in ((32, 10, 4), (16, 20, max(1, (col.numel() // n) // 16))):
        _check_gray_device_ordering(ops, oracle, rp, col, n, params, idx64)


def test_gray_stable_device_ordering_through_cpp(oracle, tmp_path):
    """GrayReorderParams::stable_device_ordering through the C++ layer (reorder_cli gray --device --stable): the same
    order as the C ABI's, on the power-law mid-size instance and on a banded one."""
    import subprocess
    from test_gpu_parity import _gray_stable_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "sparsebase_amd", "host", "bin", "reorder_cli")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(root, "sparsebase_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    for (rp, col), params in ((synth.rmat_symmetric(17, 10, seed=3), (32, 10, 4)),
                              (synth.banded_symmetric(1 << 18, 4096, 12, 5), (16, 20, 2))):
        n = len(rp) - 1
        a, b, o = (str(tmp_path / x) for x in ("rp.bin", "col.bin", "out.bin"))
        rp.tofile(a)
        col.tofile(b)
        for where in (["--device"], []):  # device-resident HIPCSR, and a host CSR staged through the GPU
            subprocess.run([cli, "gray", a, b, o, str(n), str(n), *map(str, params), *where, "--stable"], check=True,
                           env=env, timeout=600)
            got = np.fromfile(o, np.int32).astype(np.int64)
            deg, key, counts = oracle.gray_row_keys(rp, col, n, params[0], params[1])
            model, _ = _gray_stable_model(deg, key, counts, min(params[0], n), params[1], params[2])
            assert np.array_equal(got, model), (params, where)


def test_c4_one_of_eight_shards_of_1b_nnz(ops, oracle):
    """BASELINE config 4's per-rank workload on this one GPU: the scale-25 RMAT instance (~1.18 B nnz, 33.5 M rows),
    a seeded random permutation, one of the 8 new-row shards through sbx_permute_csr_rows — the properties of
    test_c3_permute_100m_properties on the shard plus the consistency of the 8 shard sizes / offsets.  (The stitch of
    row_ptr over the communicator is tests/test_sharded_gpu.py's; RCCL across 8 GPUs is bench.py --gpus 8's.)"""
    from sparsebase_amd import sharded
    rp, col = synth.rmat_symmetric_torch(25, 18, seed=1)
    n, nnz = rp.numel() - 1, col.numel()
    assert nnz > 1_000_000_000 and n == 1 << 25
    val = (torch.arange(nnz, device="cuda", dtype=torch.int32) % 1021).to(torch.float32)
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).to(torch.int32)
    ranges = sharded.row_ranges(n, 8)
    sizes = [ops.permute_csr_rows_nnz(n, rp, perm, a, b) for a, b in ranges]
    assert sum(sizes) == nnz
    deg_old = (rp[1:] - rp[:-1])
    deg_new = torch.empty_like(deg_old)
    deg_new[perm.long()] = deg_old                       # length of every new row
    offsets = torch.zeros(9, dtype=torch.int64)
    offsets[1:] = torch.cumsum(torch.tensor(sizes), 0)
    for r in range(8):  # every rank's slab size = the entries of its new rows
        a, b = ranges[r]
        assert int(deg_new[a:b].long().sum()) == sizes[r]
    # the nnz-balanced ranges of the device (what bench.py --gpus 8 and Permute2DSharded use): within one hub row of 1/8
    bal = sharded.balanced_row_ranges_device(n, rp, perm, 8)
    bal_sizes = [ops.permute_csr_rows_nnz(n, rp, perm, a, b) for a, b in bal]
    assert sum(bal_sizes) == nnz and bal[0][0] == 0 and bal[-1][1] == n
    assert max(bal_sizes) - min(bal_sizes) <= 2 * int(deg_old.max())
    # (SBX_C4_ALL_SHARDS=1: all eight shards bit-exactly, ~20 s of host time each; recorded in DESIGN.md §6)
    for r in (range(8) if os.environ.get("SBX_C4_ALL_SHARDS") else (3,)):
        a, b = ranges[r]
        srp, scol, sval = ops.permute_csr_rows(n, n, rp, col, val, perm, perm, a, b)
        k = sizes[r]
        assert scol.numel() == k and int(srp[0]) == 0 and int(srp[-1]) == k
        assert torch.equal((srp[1:] - srp[:-1]), deg_new[a:b])                 # row lengths moved with the rows
        assert ops.csr_rows_sorted(srp, scol)                                  # every row of the shard sorted
        # the shard's entries are exactly the relabelled entries of its old rows: column and payload checksums
        in_shard = (perm >= a) & (perm < b)                                    # per old row
        entry_in = torch.repeat_interleave(in_shard, deg_old.long())           # per old entry
        src_cols = col[entry_in]
        assert src_cols.numel() == k
        assert int(scol.long().sum()) == int(perm[src_cols.long()].long().sum())
        assert float(sval.double().sum()) == float(val[entry_in].double().sum())
        del entry_in, src_cols
        # global row_ptr of the shard's rows = rebased row_ptr + the shard's offset
        want_rp = torch.zeros(b - a + 1, dtype=torch.int64, device="cuda")
        want_rp[1:] = torch.cumsum(deg_new[a:b].long(), 0)
        assert torch.equal(srp.long(), want_rp) and int(offsets[r]) == int(deg_new[:a].long().sum())
        # row-wise only (the segmented copy): same lengths, every row still sorted
        rrp, rcol, _ = ops.permute_csr_rows(n, n, rp, col, val, perm, None, a, b)
        assert torch.equal(rrp, srp) and ops.csr_rows_sorted(rrp, rcol)
        del rrp, rcol
        # BIT-EXACT: the old rows that map into the shard as a CSR of their own, permuted by the oracle
        in_shard = (perm >= a) & (perm < b)
        sub_deg = deg_old[in_shard].long()
        sub_rp = np.zeros(b - a + 1, np.int32)
        sub_rp[1:] = torch.cumsum(sub_deg, 0).cpu().numpy()
        entry_in = torch.repeat_interleave(in_shard, deg_old.long())
        sub_col, sub_val = col[entry_in].cpu().numpy(), val[entry_in].cpu().numpy()
        sub_order = (perm[in_shard] - a).cpu().numpy()
        del entry_in
        want = oracle.permute_csr(sub_rp, sub_col, sub_val, sub_order, perm.cpu().numpy())
        assert np.array_equal(srp.cpu().numpy(), want[0])
        assert np.array_equal(scol.cpu().numpy(), want[1])
        assert np.array_equal(sval.cpu().numpy(), want[2])


def test_int64_permute_column_map_beyond_2_31(ops, oracle):
    """Permute2D of 64-bit arrays under a column map whose ids need more than 31 bits (m = 3 * 2^30 + 1: the map alone is
    26 GB of HBM): the wide path of sbx_permute.hip — keys (new row << 32 | new column) through the device radix sort —
    against the oracle fed with the relabelled columns (it sorts the rows like the reference's CSR constructor,
    format/csr.cc:99-157).  Also one shard of it, and C2-sized 64-bit COO records with columns up to 2^33 through the
    packed-key constructor sort (format/coo.cc:96-157)."""
    g = np.random.default_rng(23)
    m = 3 * (1 << 30) + 1
    co = torch.arange(m, dtype=torch.int64, device="cuda").mul_(5).add_(7).remainder_(m)   # 5 does not divide m: a bijection
    lens = np.concatenate([g.integers(0, 300, 3000), [9000, 20000, 129, 4096]]).astype(np.int64)
    n = len(lens)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    col = np.concatenate([np.sort(g.choice(m, l, replace=False)) for l in lens]).astype(np.int64)
    col[rp[5]:rp[5] + 2] = col[rp[5] + 2]                                           # duplicates: value order
    val = g.integers(-50, 50, len(col)).astype(np.float64)
    order = synth.random_permutation(n, 9, np.int64)
    d = lambda a: torch.from_numpy(a).cuda()
    relabelled = co[d(col)].cpu().numpy()
    assert relabelled.max() >= 1 << 31
    want = oracle.permute_csr(rp, relabelled, val, order, None, m=m)
    got = ops.permute_csr(n, m, d(rp), d(col), d(val), d(order), co)
    for a, b in zip(got, want):
        assert np.array_equal(a.cpu().numpy(), b)
    srp, scol, sval = ops.permute_csr_rows(n, m, d(rp), d(col), d(val), d(order), co, n // 3, n // 2)
    lo, hi = want[0][n // 3], want[0][n // 2]
    assert np.array_equal(srp.cpu().numpy(), want[0][n // 3:n // 2 + 1] - lo)
    assert np.array_equal(scol.cpu().numpy(), want[1][lo:hi]) and np.array_equal(sval.cpu().numpy(), want[2][lo:hi])
    del co
    # the COO constructor's sort: 10 M records, 20 row bits + 34 column bits (plain digit passes on the packed key) and
    # 20 + 11 (the hybrid: digit passes over the row's leading bits, groups sorted in LDS)
    for mm in (1 << 34, 1 << 11):
        nn, nnz = 1 << 20, 10_000_000
        gen = torch.Generator(device="cuda").manual_seed(5)
        row = torch.randint(0, nn, (nnz,), device="cuda", generator=gen, dtype=torch.int64)
        colr = torch.randint(0, mm, (nnz,), device="cuda", generator=gen, dtype=torch.int64)
        v = torch.arange(nnz, device="cuda", dtype=torch.float32)
        key, perm = torch.sort(row * mm + colr, stable=True)
        r2, c2, v2 = row.clone(), colr.clone(), v.clone()
        ops.coo_sort_(nn, mm, r2, c2, v2)
        assert torch.equal(r2, key // mm) and torch.equal(c2, key % mm)
        # equal coordinates keep no particular order in the reference (std::sort on (row, col) only): compare as sets
        assert torch.equal(torch.sort(v2)[0], v) and torch.equal(row[v2.long()], r2) and torch.equal(colr[v2.long()], c2)


def test_mixed_width_coo_csr_beyond_2_31_nonzeros(ops):
    """SBX_I32_N64 where it matters: 32-bit ids, more nonzeros than an int counts (2^31 + 2^20; 2^20 rows of 2049
    entries).  COO -> CSR writes 64-bit offsets natively (copy and move), CSR -> COO reads them; the degree order and
    the degree features take the same row_ptr.  (The permutes, sorts, RCM and Gray keep 32-bit offsets inside and refuse
    nnz >= 2^31: tests/test_gpu_parity.py::test_mixed_width_tuple_int32_ids_int64_offsets.)"""
    n, per = 1 << 20, 2049
    nnz = n * per
    assert nnz > (1 << 31)
    idx = torch.arange(nnz, dtype=torch.int64, device="cuda")
    row = torch.div(idx, per, rounding_mode="floor").to(torch.int32)
    col = torch.remainder(idx, per).to(torch.int32)
    del idx
    want_rp = torch.arange(n + 1, dtype=torch.int64, device="cuda") * per
    rp = ops.coo_to_csr(n, per, row, None, None, move=True, rows_sorted=True, offset_dtype=torch.int64)[0]
    assert rp.dtype == torch.int64 and torch.equal(rp, want_rp)
    rp2, co, _ = ops.coo_to_csr(n, per, row, col, None, offset_dtype=torch.int64)
    assert torch.equal(rp2, want_rp) and torch.equal(co, col)
    del co, rp2
    back = ops.csr_to_coo(n, per, rp, col, None, move=True)[0]
    assert back.dtype == torch.int32 and torch.equal(back, row)
    del back, row
    deg = ops.csr_degrees(rp, id_dtype=torch.int32)
    assert deg.dtype == torch.int32 and int(deg.min()) == per and int(deg.max()) == per
    inv = ops.degree_reorder(rp, True, id_dtype=torch.int32)   # equal degrees: descending id inside the degree
    assert torch.equal(inv, torch.arange(n - 1, -1, -1, dtype=torch.int32, device="cuda"))
