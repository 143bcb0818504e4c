#!/usr/bin/env python3
"""bench.py — RCM reorder + CSR permute on a 100M-nnz power-law (RMAT) CSR (BASELINE config 3).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = RCMReorder::GetReorder on the device-resident CSR followed by
Permute2D(order, order) (the canonical reorder->permute pipeline,
experiment/experiment_helper.h:81-97), inputs already in HBM.  value = rows
processed by all ranks / time (Mrows/s).  RCM's BFS is a global sequential
dependency and stays single-GPU (north_star), so with N ranks every rank runs the
pipeline on its own RMAT instance (independent objects, weak scaling, no
data-path collective); the row-range sharded permutation apply with its RCCL
all-gather of row_ptr — the step that does shard — is timed as well and reported
in "permute_apply" on the same JSON line.

Extra objects on the line:
  roofline      dominant kernel, algorithmic bytes / HIP-event time measured live
                through the library's per-kernel event hooks over the timed steps
  cpu_baseline  the real reference (oracle/_ref) or the oracle port timed on the
                host cores, rank 0 at N=1 only, on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scale", type=int, default=22, help="RMAT scale (2^scale rows)")
    ap.add_argument("--edge-factor", type=int, default=13, help="~100M nnz after symmetrisation at scale 22")
    ap.add_argument("--cpu-scale", type=int, default=22, help="RMAT scale of the CPU-baseline sample (default: the bench matrix itself)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sharded", action="store_true")
    ap.add_argument("--no-cpp", action="store_true",
                    help="skip the legs that run the C++ host layer's reorder_cli in a child process (Gray end to end, "
                         "the Reorder -> Permute2D -> Convert pipeline)")
    ap.add_argument("--leg-timeout", type=float, default=300.0,
                    help="multi-rank runs: seconds the sharded legs may take before the headline is printed without them")
    return ap.parse_args()


def kernel_alg_bytes(name, n, nnz, stats, launches):
    """ALGORITHMIC bytes moved by all launches of one kernel group during one step
    (DESIGN.md "Algorithmic bytes").  I = Z = V = 4 bytes.  The permute kernels declare the bytes of the
    nonzeros they themselves process at their launch sites (sbx_profile_query_bytes): None here."""
    if name in ("bfs_expand", "bfs_heavy"):
        # top-down: every scanned adjacency entry once (4 B) and, per frontier vertex,
        # row_ptr (8 B) + parent-position word read+write (8 B)
        td = stats["edges_scanned"] - stats["edges_scanned_bottom_up"]
        return 4 * td + 16 * stats["largest_component"] * stats["bfs_sweeps"]
    if name == "bfs_bottom_up":
        # bottom-up: scanned adjacency entries (4 B each) + the visited bitmap (n/8 B) per launch
        return 4 * stats["edges_scanned_bottom_up"] + int(launches * n / 8)
    return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # SBX_BENCH_BACKEND=gloo lets the multi-rank code path be exercised on a box with fewer GPUs than ranks
        # (ranks then share devices); the real runs use nccl = RCCL, one GPU per rank
        backend = os.environ.get("SBX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
            local_rank = local_rank % torch.cuda.device_count()
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from sparsebase_amd import ops, synth

    # ---- synthetic input: one RMAT instance per rank (seeded by rank), resident in HBM
    rp, col = synth.rmat_symmetric_torch(args.scale, args.edge_factor, seed=1 + rank, device=dev)
    n, nnz = rp.numel() - 1, col.numel()
    val = (torch.arange(nnz, device=dev, dtype=torch.int32) % 1021).to(torch.float32)
    order = torch.empty(n, dtype=torch.int32, device=dev)
    out = (torch.empty_like(rp), torch.empty_like(col), torch.empty_like(val))
    torch.cuda.synchronize()

    def step():
        ops.rcm_reorder(rp, col, out=order)
        ops.permute_csr(n, n, rp, col, val, order, order, out=out)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # ---- timed region: EXACTLY args.steps steps between two barrier + synchronize pairs.  No instrumentation
    # inside it: bracketing each of the ~330 launches of a step with HIP events costs ~1.5 ms per step
    # (measured: 11.6 vs 10.05 ms), which would be charged to the metric.
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    wall = time.perf_counter() - t0
    # ---- instrumented region: the same loop again, every kernel launch bracketed by HIP events on the
    # library's stream (sbx_profile_*), for the per-kernel roofline figures
    ops.profile_enable(True, dev)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    wall_events = time.perf_counter() - t1
    prof = ops.profile_report(dev)
    ops.profile_enable(False, dev)
    if world > 1:
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    value = n * world * args.steps / wall / 1e6  # Mrows/s, whole job

    # statistics of one RCM call (levels / sweeps / edges) for the roofline arithmetic
    _, stats = ops.rcm_reorder(rp, col, out=order, return_stats=True)

    # ---- dominant kernel + roofline (live HIP events recorded by the library on its stream)
    roofline = None
    if prof:
        per_step = {k: (ms / args.steps, cnt / args.steps) for k, (ms, cnt, _) in prof.items()}
        declared = {k: nb / args.steps for k, (_, _, nb) in prof.items() if nb}
        dom = max(per_step, key=lambda k: per_step[k][0])
        ms_step, launches_step = per_step[dom]
        alg = kernel_alg_bytes(dom, n, nnz, stats, launches_step)
        if alg is None:
            alg = declared.get(dom)   # groups whose record counts only the library knows (radix passes)
        # HBM bytes per launch from the PMC counters: NOT measured in this run (counters need their own rocprofv3 passes) but
        # read from the committed profile of the same command — traffic_source says which collection that was
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj.get(dom, {}).get("hbm_bytes_per_launch")
            meta = tj.get("_meta", {})
            traffic_source = ("profiles/pmc_traffic.json (collected %s, build %s; tools/collect_profiles.sh)"
                              % (meta.get("collected", "?"), meta.get("build", "?")))
        roofline = {
            "bound": "hbm", "kernel": dom,
            "achieved": None if alg is None else alg / launches_step / (ms_step / launches_step) / 1e6,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": None if alg is None else alg / ms_step / 1e6 / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_source,
            "avg_launch_ms": ms_step / launches_step, "launches_per_step": launches_step,
            "measured": f"HIP events around every launch over {args.steps} further steps of the same loop "
                        f"({wall_events / args.steps * 1e3:.2f} ms/step with the events in place; the permute's "
                        f"independent paths run back to back there and overlap on side streams in the timed region)",
            "alg_bytes_per_step": alg,
            "kernel_ms_per_step": {k: round(v[0], 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1][0])},
        }
        # whole operations (SURVEY §8d figures), timed un-instrumented below
        roofline["op"] = op_fractions(ops, n, nnz, rp, col, val, order, out, stats, args.steps)
        if not args.no_cpp:  # (after every timed leg of this process: the child shares the GPU)
            roofline["op"].update(cpp_legs(n, nnz, rp, col, roofline["op"]))

    extra = {"permute_apply": None, "convert_apply": None, "cpu_baseline": None}

    import threading
    emit_lock = threading.Lock()
    emitted = [False]

    def emit():
        # (the watchdog thread and the main thread may both get here: the line is printed once)
        with emit_lock:
            if emitted[0]:
                return
            emitted[0] = True
        if rank != 0:
            return
        line = {
            "metric": "Mrows/s, RCM reorder + CSR permute, 100M-nnz power-law (RMAT) CSR",
            "value": value, "unit": "Mrows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"RCMReorder + Permute2D(order,order) on symmetric RMAT scale {args.scale} "
                                   f"(a,b,c=.57,.19,.19, edge factor {args.edge_factor}) CSR <int32,int32,float32>, "
                                   "one instance per GPU",
                       "rows": n, "nnz": nnz, "rcm": stats},
            "roofline": roofline, "cpu_baseline": extra["cpu_baseline"], "permute_apply": extra["permute_apply"],
            "convert_apply": extra["convert_apply"],
        }
        cb = extra["cpu_baseline"]
        if cb and "parity_on_bench_matrix" in cb:
            line["parity_on_bench_matrix"] = cb["parity_on_bench_matrix"]
        print(json.dumps(line), flush=True)

    # The two legs below are extras next to the headline (measured above) and, with several ranks, the only part of
    # this program that depends on a collective: a leg that raises is reported as {"error": ...}, and one that hangs
    # (a rank lost inside RCCL) ends the program from a watchdog thread after the headline line has been printed —
    # the main thread may be inside a C call that no Python signal handler interrupts.
    watchdog = None
    if world > 1 and not args.no_sharded:
        def give_up():
            for k in ("permute_apply", "convert_apply"):
                if extra[k] is None:
                    extra[k] = {"error": f"no result after {args.leg_timeout} s (watchdog)"}
            emit()
            sys.stdout.flush()
            os._exit(3)  # a rank lost in a collective is a failed run: the launcher must see it

        watchdog = threading.Timer(args.leg_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()

    def guarded(name, leg):
        try:
            extra[name] = leg()
        except Exception as e:  # noqa: BLE001
            extra[name] = {"error": f"{type(e).__name__}: {e}"[:400]}
            print(f"[bench] rank {rank}: {name} failed: {extra[name]['error']}", file=sys.stderr, flush=True)

    # ---- the step that shards: row-range permutation apply + RCCL all-gather of row_ptr
    comms = []

    def leg_permute():
        from sparsebase_amd import sharded
        perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(7)).to(torch.int32)
        if world > 1:
            dist.broadcast(perm, 0)
            # each rank holds its own instance for the headline; for the sharded apply all ranks must hold the SAME
            # matrix -> the other ranks regenerate seed 1
            if rank != 0:
                rp_s, col_s = synth.rmat_symmetric_torch(args.scale, args.edge_factor, seed=1, device=dev)
            else:
                rp_s, col_s = rp, col
            val_s = (torch.arange(col_s.numel(), device=dev, dtype=torch.int32) % 1021).to(torch.float32)
        else:
            rp_s, col_s, val_s = rp, col, val
        n_s, nnz_s = rp_s.numel() - 1, col_s.numel()
        # new-row ranges of equal ENTRY counts (a power-law matrix in equal row ranges leaves one rank with several
        # times the work): computed on the device, the same on every rank (replicated input)
        ranges = sharded.balanced_row_ranges_device(n_s, rp_s, perm, world) if world > 1 else sharded.row_ranges(n_s, world)
        comm = None
        if world > 1:  # RCCL behind the C ABI (nccl backend)
            comm = sharded.make_comm(local_rank)
            comms.append(comm)
        out_s = None
        if world > 1:  # the rank's slab, sized by the shard's entries and allocated once, outside the timed loop
            cap = ops.permute_csr_rows_nnz(n_s, rp_s, perm, *ranges[rank])
            out_s = (torch.empty(n_s + 1, dtype=rp_s.dtype, device=dev), torch.empty(cap, dtype=col_s.dtype, device=dev),
                     torch.empty(cap, dtype=val_s.dtype, device=dev))

        def sharded_step():
            if world > 1:
                return ops.permute_csr_sharded(comm, n_s, n_s, rp_s, col_s, val_s, perm, perm, ranges=ranges, out=out_s)
            return ops.permute_csr(n_s, n_s, rp_s, col_s, val_s, perm, perm, out=out)

        sharded_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sharded_step()
        barrier()
        w = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([w], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w = float(t.item())
        alg = 16 * nnz_s + 12 * n_s + 8
        rowwise = None
        if world == 1:  # the row-wise variant (no column relabel gather, rows stay ordered): a pure gather
            ops.permute_csr(n_s, n_s, rp_s, col_s, val_s, perm, None, out=out)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                ops.permute_csr(n_s, n_s, rp_s, col_s, val_s, perm, None, out=out)
            torch.cuda.synchronize()
            w1 = time.perf_counter() - t1
            rowwise = {"workload": "Permute2DRowWise(random order), same matrix", "value": n_s * args.steps / w1 / 1e6,
                       "unit": "Mrows/s", "ms_per_step": w1 / args.steps * 1e3, "alg_gbs": alg * args.steps / w1 / 1e9,
                       "frac_of_hbm_peak": alg * args.steps / w1 / 1e9 / HBM_PEAK_GBS}
        return {
            "workload": f"Permute2D(random order) of one RMAT scale-{args.scale} CSR, new-row ranges over {world} GPU(s), "
                        "all-gather of row_ptr" if world > 1 else f"Permute2D(random order), RMAT scale-{args.scale}, 1 GPU",
            "scaling": "strong", "value": n_s * args.steps / w / 1e6, "unit": "Mrows/s", "ms_per_step": w / args.steps * 1e3,
            "alg_gbs": alg * args.steps / w / 1e9, "frac_of_hbm_peak": alg * args.steps / w / 1e9 / (HBM_PEAK_GBS * world),
            "rowwise": rowwise, "ranges": "nnz-balanced" if world > 1 else "whole matrix",
        }

    # ---- the other step that shards (north_star): COO -> CSR by row range, same row_ptr stitch
    def leg_convert():
        from sparsebase_amd import sharded
        if world > 1:
            rp_c, col_c = (rp, col) if rank == 0 else synth.rmat_symmetric_torch(args.scale, args.edge_factor, seed=1, device=dev)
        else:
            rp_c, col_c = rp, col
        n_c, nnz_c = rp_c.numel() - 1, col_c.numel()
        val_c = torch.ones(nnz_c, device=dev, dtype=torch.float32)
        row_c = ops.csr_to_coo(n_c, n_c, rp_c, col_c, None, move=True)[0]   # replicated row-sorted COO
        out_c = None
        comm = None
        if world > 1:
            if not comms:
                comms.append(sharded.make_comm(local_rank))
            comm = comms[0]
            ranges_c = sharded.balanced_row_ranges_device(n_c, rp_c, None, world)
            lo_c, hi_c = ranges_c[rank]
            cap = int(rp_c[hi_c] - rp_c[lo_c])
            out_c = (torch.empty(n_c + 1, dtype=rp_c.dtype, device=dev), torch.empty(cap, dtype=col_c.dtype, device=dev),
                     torch.empty(cap, dtype=val_c.dtype, device=dev))

        def convert_step():
            if world > 1:
                return ops.coo_to_csr_sharded(comm, n_c, n_c, row_c, col_c, val_c, ranges=ranges_c, out=out_c)
            return ops.coo_to_csr(n_c, n_c, row_c, col_c, val_c, rows_sorted=True)

        convert_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            convert_step()
        barrier()
        wc = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([wc], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wc = float(t.item())
        alg_c = 20 * nnz_c + 4 * (n_c + 1)
        return {
            "workload": (f"COO->CSR of one RMAT scale-{args.scale} COO, row ranges over {world} GPU(s), all-gather of row_ptr"
                         if world > 1 else f"COO->CSR (copy), RMAT scale-{args.scale}, 1 GPU"),
            "scaling": "strong", "value": n_c * args.steps / wc / 1e6, "unit": "Mrows/s", "ms_per_step": wc / args.steps * 1e3,
            "alg_gbs": alg_c * args.steps / wc / 1e9, "frac_of_hbm_peak": alg_c * args.steps / wc / 1e9 / (HBM_PEAK_GBS * world)}

    if not args.no_sharded:
        guarded("permute_apply", leg_permute)
        guarded("convert_apply", leg_convert)
    if watchdog is not None:
        watchdog.cancel()
    for c in comms:
        try:
            c.close()
        except Exception:  # noqa: BLE001
            pass

    # ---- CPU baseline: rank 0, N=1 only, bounded sample of the same workload
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        step()
        torch.cuda.synchronize()
        extra["cpu_baseline"] = run_cpu_baseline(args, synth, rp, col, gpu_result=(order, out, val))

    emit()
    if world > 1:
        dist.destroy_process_group()


def op_fractions(ops, n, nnz, rp, col, val, order, out, stats, steps):
    """Whole-operation roofline fractions on the bench matrix: Permute2D(order, order) with the RCM order
    (16 N + 12 n + 8 bytes) and RCMReorder (B (4 N + 16 n) bytes, B = the sweeps the reference's algorithm
    prescribes on this instance: pseudo-peripheral iterations + 1)."""
    def timed(f):
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps
    t_perm = timed(lambda: ops.permute_csr(n, n, rp, col, val, order, order, out=out))
    t_rcm = timed(lambda: ops.rcm_reorder(rp, col, out=torch.empty_like(order)))
    b_ref = int(stats.get("reference_sweeps") or 0) or None
    alg_p = 16 * nnz + 12 * n + 8
    res = {"permute2d": {"ms": t_perm * 1e3, "alg_bytes": alg_p, "alg_gbs": alg_p / t_perm / 1e9,
                         "frac_of_hbm_peak": alg_p / t_perm / 1e9 / HBM_PEAK_GBS}}
    if b_ref:
        alg_r = b_ref * (4 * nnz + 16 * n)
        res["rcm"] = {"ms": t_rcm * 1e3, "reference_sweeps": b_ref, "executed_sweeps": stats.get("bfs_sweeps"),
                      "unordered_sweeps": stats.get("unordered_sweeps"), "alg_bytes": alg_r, "alg_gbs": alg_r / t_rcm / 1e9,
                      "frac_of_hbm_peak": alg_r / t_rcm / 1e9 / HBM_PEAK_GBS}
    else:
        res["rcm"] = {"ms": t_rcm * 1e3}
    res["gray"] = gray_fractions(ops, n, nnz, rp, col, steps)
    return res


def gray_fractions(ops, n, nnz, rp, col, steps):
    """GrayReorder's device key stage (4 N + 16 n bytes) on the bench matrix; the whole reorderer through the C++ host
    layer is cpp_legs' business.  (BitSize32, 10, 4): the reference's test parameters,
    tests/suites/sparsebase/preprocess/preprocess_tests.cc."""
    res_, thr, grp = 32, 10, 4
    ops.gray_row_keys(n, rp, col, res_, thr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ops.gray_row_keys(n, rp, col, res_, thr)
    torch.cuda.synchronize()
    t_keys = (time.perf_counter() - t0) / steps
    alg = 4 * nnz + 16 * n
    out = {"params": [res_, thr, grp],
           "key_stage": {"ms": t_keys * 1e3, "alg_bytes": alg, "frac_of_hbm_peak": alg / t_keys / 1e9 / HBM_PEAK_GBS}}
    try:  # the opt-in ordering on the device (sbx_gray_reorder, stable ties): key stage + one radix sort of composite keys
        ops.gray_reorder(n, rp, col, res_, thr, grp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ops.gray_reorder(n, rp, col, res_, thr, grp)
        torch.cuda.synchronize()
        out["device_ordering_stable_ties"] = {"ms": (time.perf_counter() - t0) / steps * 1e3,
                                              "note": "opt-in: equals the exact mode wherever the reference's comparators "
                                                      "decide a row's place; the default (exact) mode is end_to_end"}
    except Exception as e:  # noqa: BLE001 — an extra
        out["device_ordering_stable_ties"] = {"error": repr(e)[:200]}
    return out


def cpp_legs(n, nnz, rp, col, op):
    """What a C++ caller of the boundary gets, measured in a child process that runs the host layer's reorder_cli on the
    bench matrix (device-resident HIPCSR, warm calls): the canonical pipeline of the reference's experiment helper
    (experiment/experiment_helper.h:81-97: Reorder<RCMReorder> -> Permute2D -> Convert) next to this process's `ops`
    figures, and GrayReorder end to end (device key stage, degrees and keys to the host, the host ordering stage).  The
    pipeline is timed twice: with the device-resident overloads (the order vector stays in HBM as an HIPArray) and, under
    `host_order_vector`, with the reference's own signatures (the order crosses PCIe twice: returned and taken as a host array)."""
    import subprocess
    import tempfile
    res = {}
    cli = os.path.join(ROOT, "sparsebase_amd", "host", "bin", "reorder_cli")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "sparsebase_amd", "lib") + ":" +
               os.environ.get("LD_LIBRARY_PATH", ""))
    try:
        if not os.path.exists(cli):
            raise RuntimeError("sparsebase_amd/host/bin/reorder_cli is not built")
        with tempfile.TemporaryDirectory() as tmp:
            a, b, o = (os.path.join(tmp, x) for x in ("rp.bin", "col.bin", "out.bin"))
            rp.cpu().numpy().tofile(a)
            col.cpu().numpy().tofile(b)
            try:
                r = subprocess.run([cli, "pipeline", a, b, o, str(n), str(n)], env=env, capture_output=True, text=True,
                                   timeout=600, check=True)
                lines = r.stdout.split("\n")
                # line 1: the reference's own signatures (the order vector is a host array between Reorder and Permute2D);
                # line 2: the device-resident overloads (Reorder -> HIPArray<int> -> Permute2D), what `ops` measures too
                h_re, h_pe, h_co = (float(x) for x in lines[0].split())
                t_re, t_pe, t_co = (float(x) for x in lines[1].split())
                pipe = {"reorder_ms": t_re, "permute2d_ms": t_pe, "convert_ms": t_co, "ms": t_re + t_pe + t_co,
                        "via": "host/bin/reorder_cli pipeline (HIPCSR<int,int,float>, best of five warm rounds; the order "
                               "vector stays in HBM: ReorderBase::Reorder(params, format, HIPContext&) -> HIPArray -> Permute2D)",
                        "host_order_vector": {"reorder_ms": h_re, "permute2d_ms": h_pe, "convert_ms": h_co,
                                              "ms": h_re + h_pe + h_co,
                                              "order_vector_over_pcie_mb": round(2 * 4 * n / 1e6, 1),
                                              "note": "the reference's signatures: Reorder returns a host IDType*, "
                                                      "Permute2D takes one"}}
                if "rcm" in op and "permute2d" in op:
                    pipe["ops_ms"] = op["rcm"]["ms"] + op["permute2d"]["ms"]
                    pipe["over_ops"] = pipe["ms"] / pipe["ops_ms"]
                    pipe["host_order_vector"]["over_ops"] = pipe["host_order_vector"]["ms"] / pipe["ops_ms"]
                res["pipeline_cpp"] = pipe
            except Exception as e:  # noqa: BLE001 — an extra: the line is still the line without it
                res["pipeline_cpp"] = {"error": repr(e)[:200]}
            try:
                res_, thr, grp = op["gray"]["params"]
                r = subprocess.run([cli, "gray", a, b, o, str(n), str(n), str(res_), str(thr), str(grp), "--device", "--time",
                                    "--reps", "5"],
                                   env=env, capture_output=True, text=True, timeout=600, check=True)
                lines = r.stdout.split("\n")
                dev_ms, d2h_ms, host_ms = (float(x) for x in lines[1].split())
                calls = [l for l in r.stderr.splitlines() if l.startswith("timed calls (ms):")]
                res["gray"] = dict(op["gray"], end_to_end={
                    "ms": float(lines[0]) * 1e3, "device_key_stage_ms": dev_ms, "keys_to_host_ms": d2h_ms,
                    "host_ordering_ms": host_ms,
                    "via": "host/bin/reorder_cli --device --time --reps 5 (the call of median duration of five warm calls)",
                    "calls_ms": [float(x) for x in calls[-1].split(":")[1].split()] if calls else None,
                    "note": "degrees and keys are downloaded into page-locked, pooled host blocks: with pageable targets "
                            "(rounds 3 - 5) the runtime's unpinning of the previous call's targets held this call's first "
                            "kernel back by 14 - 25 ms (tools/gray_kt2.sh, tools/gray_stall_probe.sh)"})
            except Exception as e:  # noqa: BLE001
                res["gray"] = dict(op["gray"], end_to_end={"error": repr(e)[:200]})
    except Exception as e:  # noqa: BLE001
        res["pipeline_cpp"] = {"error": repr(e)[:200]}
    return res


def run_cpu_baseline(args, synth, rp_dev, col_dev, gpu_result=None):
    """Times the reference pipeline on the host: real reference if oracle/_ref is present
    ("reference"), else the oracle restatement ("port").  Default sample: rank 0's bench matrix itself,
    one repetition (≈ 10 s of host work); --cpu-scale selects a smaller RMAT instance."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    if args.cpu_scale == args.scale:
        rp_t, col_t = rp_dev, col_dev
    else:
        rp_t, col_t = synth.rmat_symmetric_torch(args.cpu_scale, args.edge_factor, seed=1)
    rp, col = rp_t.cpu().numpy(), col_t.cpu().numpy()
    n, nnz = len(rp) - 1, len(col)
    val = (np.arange(nnz) % 1021).astype(np.float32)
    if orc.ref_available():
        impl, kind = orc.Ref(), "reference"
        cores = os.cpu_count()  # OpenMP default; only the CSR-ctor loops (format/csr.cc:102,123) are parallel
    else:
        impl, kind = orc.Oracle(), "port"
        cores = 1
    last = {}

    def run_once():
        t0 = time.perf_counter()
        order = impl.rcm_reorder(rp, col)
        t1 = time.perf_counter()
        if kind == "reference":
            res = impl.permute_csr(rp, col, val, order, order, m=n)
        else:
            res = impl.permute_csr(rp, col, val, order, order)
        t2 = time.perf_counter()
        last["order"], last["csr"] = order, res
        return t1 - t0, t2 - t1

    cold = run_once()
    rcm_s, permute_s = run_once()  # the warm repetition is the one reported
    out = {"value": n / (rcm_s + permute_s) / 1e6, "unit": "Mrows/s", "cores": cores, "kind": kind,
           "sample": (f"the bench matrix itself (symmetric RMAT scale {args.cpu_scale}, n={n}, nnz={nnz}), "
                      "second of 2 repetitions" if args.cpu_scale == args.scale else
                      f"same pipeline on symmetric RMAT scale {args.cpu_scale} (n={n}, nnz={nnz}), second of 2 repetitions"),
           "rcm_s": rcm_s, "permute_s": permute_s, "cold": {"rcm_s": cold[0], "permute_s": cold[1]}}
    if gpu_result is not None and args.cpu_scale == args.scale:
        # the headline configuration verified end to end: the GPU step's outputs against the CPU leg's, bit for bit
        g_order, g_out, _ = gpu_result
        ok = bool(np.array_equal(g_order.cpu().numpy(), last["order"]))
        for g, w in zip(g_out, last["csr"]):
            ok = ok and bool(np.array_equal(g.cpu().numpy(), w))
        out["parity_on_bench_matrix"] = ok
    if kind == "reference":
        # only the CSR constructor's two loops are OpenMP-parallel in the reference (format/csr.cc:102,123):
        # the same run with one thread (SURVEY §8d asks for both)
        try:
            import ctypes
            omp = ctypes.CDLL("libgomp.so.1")
            omp.omp_set_num_threads(1)
            r1, p1 = run_once()
            omp.omp_set_num_threads(int(cores))
            out["one_thread"] = {"value": n / (r1 + p1) / 1e6, "rcm_s": r1, "permute_s": p1}
        except OSError:
            pass
    return out


if __name__ == "__main__":
    main()
