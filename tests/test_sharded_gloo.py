"""world_size-2 gloo tests of the multi-GPU path (sparsebase_amd/sharded.py): row-range split,
all-gather of nnz totals and of the row_ptr segments.  The per-shard computation is injected as a
CPU test double (the oracle restricted to the shard's rows), so exactly the collective / stitching
code that runs over RCCL on the GPU box is exercised here."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A TCP port the OS hands out as free right now (parallel test runs must not meet on a fixed one)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _worker(rank, world, port, gather_entries, balanced, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from orc import Oracle
        from sparsebase_amd import sharded, synth
        orc = Oracle()
        rp, col = synth.rmat_symmetric(11, 8, seed=5)
        n = len(rp) - 1
        val = (np.arange(len(col)) % 31).astype(np.float32)
        order = synth.random_permutation(n, 9)
        want = orc.permute_csr(rp, col, val, order, order)

        def shard_fn(lo, hi):  # test double for ops.permute_csr_rows
            a, b = want[0][lo], want[0][hi]
            return (torch.from_numpy(want[0][lo:hi + 1] - a), torch.from_numpy(want[1][a:b].copy()),
                    torch.from_numpy(want[2][a:b].copy()))

        ranges = None
        if balanced:
            ranges = sharded.balanced_row_ranges(torch.from_numpy(want[0].astype(np.int64)), world)
        grp, lcol, lval, (lo, hi), offsets = sharded.permute_csr_sharded(
            n, n, None, None, None, None, None, ranges=ranges, shard_fn=shard_fn, gather_entries=gather_entries)
        ok = np.array_equal(grp.numpy(), want[0])
        if gather_entries:
            ok = ok and np.array_equal(lcol.numpy(), want[1]) and np.array_equal(lval.numpy(), want[2])
        else:
            a, b = want[0][lo], want[0][hi]
            ok = ok and np.array_equal(lcol.numpy(), want[1][a:b]) and int(offsets[rank]) == a
        q.put((rank, bool(ok), (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("gather_entries,balanced", [(False, False), (True, False), (False, True)])
def test_sharded_permute_two_ranks_gloo(gather_entries, balanced):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, gather_entries, balanced, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results), results
    ranges = sorted(r for _, _, r in results)
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0]  # contiguous cover


def _convert_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from orc import Oracle
        from sparsebase_amd import sharded, synth
        orc = Oracle()
        rp, col = synth.rmat_symmetric(11, 8, seed=6)   # many empty rows, hubs
        n = len(rp) - 1
        val = (np.arange(len(col)) % 29).astype(np.float32)
        row, col2, val2 = orc.csr_to_coo(rp, col, val)

        def coo_shard(lo, hi, a, b):  # CPU double of ops.coo_to_csr on the shard's slice
            lrp, lc, lv = orc.coo_to_csr(hi - lo, row[a:b] - lo, col2[a:b], val2[a:b])
            return torch.from_numpy(lrp), torch.from_numpy(lc), torch.from_numpy(lv)

        grp, lcol, lval, (lo, hi), offsets = sharded.coo_to_csr_sharded(
            n, n, torch.from_numpy(row), torch.from_numpy(col2), torch.from_numpy(val2), shard_fn=coo_shard)
        ok = np.array_equal(grp.numpy(), rp)
        a, b = rp[lo], rp[hi]
        ok = ok and np.array_equal(lcol.numpy(), col[a:b]) and np.array_equal(lval.numpy(), val[a:b])
        ok = ok and int(offsets[rank]) == a

        def csr_shard(lo_, hi_, a_, b_):  # CPU double of ops.csr_to_coo on the shard's rows
            r, c, v = orc.csr_to_coo((rp[lo_:hi_ + 1] - rp[lo_]).astype(np.int32), col[a_:b_], val[a_:b_])
            return torch.from_numpy(r + lo_), torch.from_numpy(c), torch.from_numpy(v)

        lrow, lc, lv, _, (a2, b2) = sharded.csr_to_coo_sharded(
            n, n, torch.from_numpy(rp), torch.from_numpy(col), torch.from_numpy(val), shard_fn=csr_shard,
            gather_entries=True)
        ok = ok and (a2, b2) == (a, b)
        ok = ok and np.array_equal(lrow.numpy(), row) and np.array_equal(lc.numpy(), col2) and np.array_equal(lv.numpy(), val2)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_sharded_conversions_two_ranks_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_convert_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in results), results


def test_row_ranges():
    from sparsebase_amd import sharded
    assert sharded.row_ranges(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert sharded.row_ranges(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    prefix = torch.tensor([0, 100, 100, 101, 102, 200])
    r = sharded.balanced_row_ranges(prefix, 2)
    assert r[0][0] == 0 and r[-1][1] == 5 and r[0][1] == r[1][0]
