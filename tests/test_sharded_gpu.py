"""The sharded steps END TO END on the GPU: two fresh child processes, both on GPU 0, gloo between them, each calling
sbx_permute_csr_sharded / sbx_coo_to_csr_sharded (the real HIP shard computation + the device-side stitch of
sparsebase_amd/csrc/sbx_sharded.hip, all-gathers through the communicator hook) — compared with the oracle's
whole-matrix result.  (RCCL itself needs one GPU per rank: bench.py --gpus N uses it on the multi-GPU node.)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A TCP port the OS hands out as free right now (parallel test runs must not meet on a fixed one)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _child(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from orc import Oracle
        from sparsebase_amd import ops, sharded, synth
        torch.cuda.set_device(0)
        orc = Oracle()
        d = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
        h = lambda t: None if t is None else t.cpu().numpy()
        rp, col = synth.rmat_symmetric(13, 8, seed=5)   # power-law rows: every sort class, shard boundaries inside hubs
        n = len(rp) - 1
        val = (np.arange(len(col)) % 31).astype(np.float32)
        order = synth.random_permutation(n, 9)
        want = orc.permute_csr(rp, col, val, order, order)
        comm = sharded.make_comm(0)
        ok, notes = True, []
        capi_bad_arg, capi_internal = 1, 6

        def check(tag, cond):
            nonlocal ok
            if not cond:
                notes.append(tag)
                ok = False

        for idt in (np.int32, np.int64):
            drp, dcol, dord = d(rp.astype(idt)), d(col.astype(idt)), d(order.astype(idt))
            for balanced in (False, True):
                ranges = sharded.balanced_row_ranges(torch.from_numpy(want[0].astype(np.int64)), world) if balanced else None
                grp, lcol, lval, (lo, hi), offs = sharded.permute_csr_sharded(n, n, drp, dcol, d(val), dord, dord,
                                                                             ranges=ranges, comm=comm)
                a, b = int(want[0][lo]), int(want[0][hi])
                check(f"perm rp {idt} {balanced}", np.array_equal(h(grp), want[0].astype(idt)))
                check(f"perm col {idt} {balanced}", np.array_equal(h(lcol), want[1][a:b].astype(idt)))
                check(f"perm val {idt} {balanced}", np.array_equal(h(lval), want[2][a:b]))
                check(f"perm off {idt} {balanced}", int(offs[rank]) == a and int(offs[-1]) == len(col))
        # every rank ends up with the whole matrix
        grp, lcol, lval, _, _ = sharded.permute_csr_sharded(n, n, d(rp), d(col), d(val), d(order), d(order), comm=comm,
                                                            gather_entries=True)
        check("perm gather", np.array_equal(h(lcol), want[1]) and np.array_equal(h(lval), want[2]))
        # pattern-only matrix, row order only (the segmented-copy path inside the shard)
        w2 = orc.permute_csr(rp, col, None, order, None)
        grp, lcol, lval, (lo, hi), _ = sharded.permute_csr_sharded(n, n, d(rp), d(col), None, d(order), None, comm=comm)
        check("rowwise", np.array_equal(h(grp), w2[0]) and np.array_equal(h(lcol), w2[1][w2[0][lo]:w2[0][hi]]) and lval is None)
        # COO -> CSR by row range (ragged ranges: one rank gets most rows, an empty range too)
        row, col2, val2 = orc.csr_to_coo(rp, col, val)
        for ranges in (None, [(0, 5), (5, n)], [(0, 0), (0, n)]):
            grp, lcol, lval, (lo, hi), offs = sharded.coo_to_csr_sharded(n, n, d(row), d(col2), d(val2), ranges=ranges,
                                                                         comm=comm)
            a, b = int(rp[lo]), int(rp[hi])
            check(f"coo rp {ranges}", np.array_equal(h(grp), rp))
            check(f"coo col {ranges}", np.array_equal(h(lcol), col[a:b]) and np.array_equal(h(lval), val[a:b]))
        # CSR -> COO by row range behind the C ABI (sbx_csr_to_coo_sharded), 32- and 64-bit indices, ragged ranges
        for idt in (np.int32, np.int64):
            for ranges in (None, [(0, 7), (7, n)], [(0, n), (n, n)]):
                lrow, lcol, lval, (lo, hi), (a, b) = sharded.csr_to_coo_sharded(n, n, d(rp.astype(idt)), d(col.astype(idt)),
                                                                                d(val), ranges=ranges, comm=comm)
                check(f"csr->coo bounds {idt} {ranges}", (a, b) == (int(rp[lo]), int(rp[hi])))
                check(f"csr->coo row {idt} {ranges}", np.array_equal(h(lrow), row[a:b].astype(idt)))
                check(f"csr->coo col {idt} {ranges}", np.array_equal(h(lcol), col2[a:b].astype(idt)) and np.array_equal(h(lval), val2[a:b]))
        # nnz-balanced ranges from the device (sbx_balanced_row_splits) = the host restatement on the oracle's row_ptr
        for idt in (np.int32, np.int64):
            for w in (2, 3, 8):
                got = sharded.balanced_row_ranges_device(n, d(rp.astype(idt)), d(order.astype(idt)), w)
                ref = sharded.balanced_row_ranges(torch.from_numpy(want[0].astype(np.int64)), w)
                check(f"balanced splits {idt} {w}: {got} vs {ref}", got == ref)
        # a rank whose slab does not fit: EVERY rank returns an error (nobody is left inside a collective)
        from sparsebase_amd import capi
        cap = ops.permute_csr_rows_nnz(n, d(rp), d(order), *sharded.row_ranges(n, world)[rank])
        small = (torch.empty(n + 1, dtype=torch.int32).cuda(), torch.empty(max(cap - (5 if rank == 1 else 0), 1), dtype=torch.int32).cuda(),
                 torch.empty(max(cap, 1), dtype=torch.float32).cuda())
        try:
            sharded.permute_csr_sharded(n, n, d(rp), d(col), d(val), d(order), d(order), comm=comm, out=small)
            check("a failing rank must fail every rank", False)
        except capi.SbxError as e:
            check(f"status of rank {rank}: {e}", e.status == (capi_bad_arg if rank == 1 else capi_internal))
        # ... and the communicator is still usable afterwards
        grp, lcol, lval, (lo, hi), offs = sharded.permute_csr_sharded(n, n, d(rp), d(col), d(val), d(order), d(order), comm=comm)
        check("after the failure", np.array_equal(h(grp), want[0]))
        comm.close()
        q.put((rank, ok, notes))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, False, [repr(e), traceback.format_exc()]))
    finally:
        dist.destroy_process_group()


def test_sharded_steps_two_ranks_one_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no GPU is visible (the HIP path has no CPU fallback)")
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")  # fresh interpreters: nothing GPU-related is inherited
    q = ctx.Queue()
    procs = [ctx.Process(target=_child, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, notes in results:
        assert ok, f"rank {rank}: {notes}"


def test_rccl_communicator_single_rank():
    """The RCCL flavour of the communicator (librccl loaded by the C ABI, ncclCommInitRank / ncclAllGather called
    from C++) on the one GPU of this box: a world of one rank, whole-matrix result compared with the oracle."""
    import ctypes as C
    from orc import Oracle
    from sparsebase_amd import capi, ops, synth
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no GPU is visible (the HIP path has no CPU fallback)")
    lib = capi.load()
    buf = C.create_string_buffer(capi.COMM_ID_BYTES)
    assert lib.sbx_comm_unique_id(buf) == capi.SBX_OK
    c = C.c_void_p()
    assert lib.sbx_comm_create_rccl(0, 0, 1, buf, C.byref(c)) == capi.SBX_OK
    comm = ops.Comm(lib, c)
    try:
        orc = Oracle()
        rp, col = synth.rmat_symmetric(12, 8, seed=3)
        n = len(rp) - 1
        val = (np.arange(len(col)) % 17).astype(np.float32)
        order = synth.random_permutation(n, 2)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        grp, lcol, lval, offs = ops.permute_csr_sharded(comm, n, n, d(rp), d(col), d(val), d(order), d(order))
        want = orc.permute_csr(rp, col, val, order, order)
        assert np.array_equal(grp.cpu().numpy(), want[0]) and np.array_equal(lcol.cpu().numpy(), want[1])
        assert np.array_equal(lval.cpu().numpy(), want[2]) and offs == [0, len(col)]
        row, col2, val2 = orc.csr_to_coo(rp, col, val)
        grp, lcol, lval, offs = ops.coo_to_csr_sharded(comm, n, n, d(row), d(col2), d(val2))
        assert np.array_equal(grp.cpu().numpy(), rp) and np.array_equal(lcol.cpu().numpy(), col)
    finally:
        comm.close()


def test_bench_eight_ranks_on_one_gpu():
    """bench.py --gpus 8 as the driver launches it (torch.distributed.run, 8 fresh processes), but over gloo with all
    ranks on GPU 0 and a small matrix: the perm broadcast, the per-rank regeneration, the nnz-balanced ranges, slab
    sizing, both sharded legs through the C ABI and the watchdog all run with 8 ranks; the line must carry both legs
    without an error."""
    import json
    import subprocess
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no GPU is visible (the HIP path has no CPU fallback)")
    env = dict(os.environ, SBX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--scale", "16", "--steps", "2",
           "--warmup", "1", "--leg-timeout", "240"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:  # the ranks' own messages sit in front of torchrun's summary
        first = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "sbx" in l][:40]
        dump = os.path.join(ROOT, "gpurun_out", "bench8_stderr.log")
        try:
            os.makedirs(os.path.dirname(dump), exist_ok=True)
            open(dump, "w").write(r.stderr)
        except OSError:
            pass
        assert False, "\n".join(first) + "\n...\n" + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["value"] > 0
    for leg in ("permute_apply", "convert_apply"):
        assert line[leg] is not None and "error" not in line[leg], line[leg]
    assert line["permute_apply"]["ranges"] == "nnz-balanced"


def test_rcm_eight_processes_on_one_gpu():
    """Eight processes ordering the same graphs on ONE GPU at once: the persistent kernels' grid barriers give up now and
    then, workgroups of one launch get far apart in time — the results must stay the oracle's.  (Round 3: a workgroup
    that was slow to read the cone list's counter behind a barrier walked a different range than the others, about one
    call in a hundred ended in "not structurally symmetric"; fixed in k_ubfs_cone_run.)"""
    import subprocess
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no GPU is visible (the HIP path has no CPU fallback)")
    env = dict(os.environ, RCM_MODES_SMALL="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "rcm_modes.py")], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(8)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-1500:]


def test_rcm_eight_processes_thousands_of_calls():
    """Eight processes, 400 rounds over four graphs each, on ONE GPU, every call trying the persistent kernels again
    (SBX_DEBUG_GB_BACKOFF=0): 12 800 orderings against the oracle.  What this guards: a workgroup's stores for the other
    workgroups — the small levels' queue entries and hub queue — have to arrive before its arrival at the grid barrier
    is counted (gb_wait waits for them; `__syncthreads()` alone does not), or a workgroup on another XCD passes the
    barrier and reads the old word.  With the memory side congested by the other processes that lost part of a level
    about once in 5000 calls — rounds 2 to 4 carried it."""
    import subprocess
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no GPU is visible (the HIP path has no CPU fallback)")
    env = dict(os.environ, SBX_DEBUG_GB_BACKOFF="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "rcm_shared_loop.py"), "400"], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(8)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "mismatches 0 exceptions 0" in o, o[-1500:]


def test_rcm_fenced_build_agrees():
    """The grid barriers of the persistent RCM kernels carry no release / acquire fence: they are sound while every word
    one workgroup hands another between two barriers moves through agent-scope atomics (sbx_rcm.hip, gb_wait).  The
    checking build libsbx_fenced.so (-DSBX_GB_FENCED: fences at every barrier and election) needs no such invariant.
    Both builds run the stress loop — four processes at once on the GPU, every call trying the persistent kernels again,
    the closure check behind every unordered sweep (SBX_DEBUG_RCM_CHECK) — and must agree: the oracle's orders in every
    call, no edge left open, identical digests."""
    import subprocess
    from sparsebase_amd import build as hip_build
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no GPU is visible (the HIP path has no CPU fallback)")
    fenced = hip_build.variant_path("fenced")
    assert os.path.exists(fenced), "libsbx_fenced.so is not built (__graft_entry__.build() builds it)"
    digests = {}
    for tag in ("product", "fenced"):
        env = dict(os.environ, SBX_DEBUG_GB_BACKOFF="0", SBX_DEBUG_RCM_CHECK="1")
        if tag == "fenced":
            env["SBX_PROBE_LIB"] = "fenced"
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "rcm_shared_loop.py"), "100", "--digests"],
                                  cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                 for _ in range(4)]
        outs = [p.communicate(timeout=900)[0] for p in procs]
        for p, o in zip(procs, outs):
            assert p.returncode == 0 and "mismatches 0 exceptions 0" in o, (tag, o[-1500:])
            assert "[rcm check]" not in o, (tag, o[-1500:])  # an unordered sweep left an edge open
            d = tuple(l for l in o.splitlines() if l.startswith("digest"))
            assert len(d) == 4, (tag, o[-800:])
            digests.setdefault(tag, set()).add(d)
    assert len(digests["product"]) == 1 and digests["product"] == digests["fenced"]
