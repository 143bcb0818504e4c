"""Row-range sharded permutation apply and format conversion across the GPUs of one node
(SURVEY.md §8e).

Decomposition: new row i depends only on old row old_of_new[i] and the replicated
col_order table, so rank r owns new rows [lo_r, hi_r).  Every rank holds the input
CSR and the order vectors (replicated); it produces its slab of col/val locally
(sbx_permute_csr_rows) and the only exchange is

  1. all_gather of the per-rank nnz totals      (world_size int64 values)
  2. all_gather of the rebased row_ptr segments (Z*n bytes in total)

after which every rank holds the complete permuted row_ptr; col/val stay
row-sharded (a distributed CSR) unless gather_entries=True.  The product path is ONE
C-ABI call per rank (sbx_permute_csr_sharded / sbx_coo_to_csr_sharded,
sparsebase_amd/csrc/sbx_sharded.hip): the rank's slab, both all-gathers (RCCL
ncclAllGather from C++ when the torch.distributed backend is nccl; an all-gather hook
over the group otherwise) and the stitch kernel.  This module only chooses the row
ranges and builds the communicator; the CPU gloo tests inject the shard computation and
run the host-language restatement of the stitch (stitch_row_ptr).
"""
import torch
import torch.distributed as dist


def row_ranges(n, world_size):
    """Contiguous new-row ranges, sizes differing by at most one row."""
    base, extra = divmod(n, world_size)
    out, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def balanced_row_ranges_device(n, row_ptr, row_order, world_size):
    """Ranges balanced by nnz instead of rows (power-law inputs), computed behind the C ABI
    (sbx_balanced_row_splits: scatter of the new rows' lengths, scan, one search per split — all on the device)."""
    from . import ops
    return ops.balanced_row_splits(n, row_ptr, row_order, world_size)


def balanced_row_ranges(new_row_lengths_prefix, world_size):
    """Host-language restatement of sbx_balanced_row_splits for the CPU gloo tests: split points
    from the scanned new-row lengths (a length n+1 CPU tensor)."""
    total = int(new_row_lengths_prefix[-1])
    n = new_row_lengths_prefix.numel() - 1
    cuts = [0]
    for r in range(1, world_size):
        target = total * r // world_size
        cuts.append(int(torch.searchsorted(new_row_lengths_prefix, torch.tensor(target), right=False)))
    cuts.append(n)
    cuts = [min(max(c, 0), n) for c in cuts]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return list(zip(cuts[:-1], cuts[1:]))


def make_comm(device_index, group=None):
    """The sbx communicator for a torch.distributed group: RCCL (ncclAllGather called from the C ABI, one GPU per
    rank) when the group's backend is nccl, otherwise the all-gather hook over the group (gloo: ranks may share a GPU)."""
    from . import ops
    if dist.get_backend(group) == "nccl":
        return ops.Comm.rccl(device_index, group)
    return ops.Comm.hook(device_index, group)


def stitch_row_ptr(local_row_ptr, ranges, group=None):
    """Host-language restatement of the stitch the C ABI does on the device (sbx_sharded.hip: nnz totals, padded
    equal chunks, one all-gather each): used with CPU tensors by the gloo tests, which inject the shard computation.

    local_row_ptr: (hi-lo+1,) tensor, rebased to 0.  Returns (global_row_ptr (n+1,), shard_offsets (world+1,) int64).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = ranges[rank]
    assert local_row_ptr.numel() == hi - lo + 1
    dev, dt = local_row_ptr.device, local_row_ptr.dtype
    totals = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(totals, local_row_ptr[-1:].to(torch.int64), group=group)
    offsets = torch.zeros(world + 1, dtype=torch.int64, device=dev)
    torch.cumsum(totals, 0, out=offsets[1:])
    starts = torch.tensor([l for l, _ in ranges] + [ranges[-1][1]], dtype=torch.int64, device=dev)
    chunk = int((starts[1:] - starts[:-1]).max().clamp(min=1))
    seg = torch.zeros(chunk, dtype=dt, device=dev)
    seg[: hi - lo] = local_row_ptr[:-1] + offsets[rank].to(dt)
    gathered = torch.empty(world * chunk, dtype=dt, device=dev)
    dist.all_gather_into_tensor(gathered, seg, group=group)
    n = ranges[-1][1]
    rows = torch.arange(n, dtype=torch.int64, device=dev)
    owner = torch.searchsorted(starts[1:], rows, right=True).clamp(max=world - 1)
    out = torch.empty(n + 1, dtype=dt, device=dev)
    out[:n] = gathered[owner * chunk + rows - starts[owner]]
    out[n] = offsets[world].to(dt)
    return out, offsets


def permute_csr_sharded(n, m, row_ptr, col, val, row_order, col_order, group=None, ranges=None,
                        shard_fn=None, gather_entries=False, comm=None, out=None):
    """Sharded Permute2D.  Returns (global_row_ptr, local_col, local_val, (lo, hi), offsets).

    Product path (shard_fn is None): one call of sbx_permute_csr_sharded — the rank's slab and both all-gathers
    happen behind the C ABI; `comm` is an ops.Comm (make_comm), `out` optional pre-allocated
    (row_ptr_out, col_out, val_out).  shard_fn(lo, hi) -> (rebased_row_ptr, col, val) is the tests' injection point:
    then the stitch runs through torch.distributed on whatever tensors the double returns."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ranges = row_ranges(n, world) if ranges is None else ranges
    lo, hi = ranges[rank]
    if shard_fn is None:
        from . import ops
        own = comm is None
        comm = make_comm(row_ptr.device.index, group) if own else comm
        try:
            grp, lcol, lval, offs = ops.permute_csr_sharded(comm, n, m, row_ptr, col, val, row_order, col_order,
                                                            ranges=ranges, out=out)
        finally:
            if own:
                comm.close()
        offsets = torch.tensor(offs, dtype=torch.int64, device=row_ptr.device)
    else:
        lrp, lcol, lval = shard_fn(lo, hi)
        grp, offsets = stitch_row_ptr(lrp, ranges, group)
    if gather_entries:
        lcol = _gather_v(lcol, offsets, group)
        lval = None if lval is None else _gather_v(lval, offsets, group)
    return grp, lcol, lval, (lo, hi), offsets


def coo_to_csr_shard(m, row, col, val, lo, hi, a, b):
    """HIP path of one COO -> CSR shard: rows [lo, hi) = nonzeros [a, b) of the row-sorted COO."""
    from . import ops
    return ops.coo_to_csr(hi - lo, m, row[a:b] - lo, col[a:b], None if val is None else val[a:b], rows_sorted=True)


def csr_to_coo_shard(m, row_ptr, col, val, lo, hi, a, b):
    """HIP path of one CSR -> COO shard (global row ids)."""
    from . import ops
    r, c, v = ops.csr_to_coo(hi - lo, m, row_ptr[lo:hi + 1] - row_ptr[lo], col[a:b], None if val is None else val[a:b])
    return r + lo, c, v


def coo_to_csr_sharded(n, m, row, col, val, group=None, ranges=None, shard_fn=None, gather_entries=False, comm=None,
                       out=None):
    """Sharded COO -> CSR of a row-sorted COO (replicated input, like the permute): rank r owns rows
    [lo_r, hi_r); its nonzeros are the contiguous slice found by binary search on the sorted row array,
    converted locally, and the row_ptr segments are stitched with the same two all-gathers — all behind
    sbx_coo_to_csr_sharded unless a shard_fn(lo, hi, a, b) -> (rebased_row_ptr, col, val) double is injected.
    Returns (global_row_ptr, local_col, local_val, (lo, hi), offsets)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ranges = row_ranges(n, world) if ranges is None else ranges
    lo, hi = ranges[rank]
    if shard_fn is None:
        from . import ops
        own = comm is None
        comm = make_comm(row.device.index, group) if own else comm
        try:
            grp, lcol, lval, offs = ops.coo_to_csr_sharded(comm, n, m, row, col, val, ranges=ranges, out=out)
        finally:
            if own:
                comm.close()
        offsets = torch.tensor(offs, dtype=torch.int64, device=row.device)
    else:
        bounds = torch.searchsorted(row, torch.tensor([lo, hi], dtype=row.dtype, device=row.device), right=False)
        a, b = int(bounds[0]), int(bounds[1])
        lrp, lcol, lval = shard_fn(lo, hi, a, b)
        grp, offsets = stitch_row_ptr(lrp, ranges, group)
    if gather_entries:
        lcol = _gather_v(lcol, offsets, group)
        lval = None if lval is None else _gather_v(lval, offsets, group)
    return grp, lcol, lval, (lo, hi), offsets


def csr_to_coo_sharded(n, m, row_ptr, col, val, group=None, ranges=None, shard_fn=None, gather_entries=False, comm=None,
                       out=None):
    """Sharded CSR -> COO: rank r expands the row ids of rows [lo_r, hi_r) (nonzeros
    [row_ptr[lo], row_ptr[hi])); no collective is needed unless the caller wants every rank to hold the
    whole COO (gather_entries).  Product path (shard_fn is None): one call of sbx_csr_to_coo_sharded.
    Returns (local_row, local_col, local_val, (lo, hi), (a, b))."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ranges = row_ranges(n, world) if ranges is None else ranges
    lo, hi = ranges[rank]
    if shard_fn is None:
        from . import ops
        own = comm is None
        comm = make_comm(row_ptr.device.index, group) if own else comm
        try:
            lrow, lcol, lval, offs = ops.csr_to_coo_sharded(comm, n, m, row_ptr, col, val, ranges=ranges, out=out)
        finally:
            if own:
                comm.close()
        a, b = offs[rank], offs[rank + 1]
    else:
        a, b = int(row_ptr[lo]), int(row_ptr[hi])
        lrow, lcol, lval = shard_fn(lo, hi, a, b)
    if gather_entries:
        cuts = torch.tensor([int(row_ptr[l]) for l, _ in ranges] + [int(row_ptr[ranges[-1][1]])], dtype=torch.int64,
                            device=row_ptr.device)
        lrow = _gather_v(lrow, cuts, group)
        lcol = _gather_v(lcol, cuts, group)
        lval = None if lval is None else _gather_v(lval, cuts, group)
    return lrow, lcol, lval, (lo, hi), (a, b)


def _gather_v(local, offsets, group=None):
    """All-gather-v of row-sharded entries (only when the caller wants every rank to hold them all)."""
    world = dist.get_world_size(group)
    sizes = offsets[1:] - offsets[:-1]
    cap = int(sizes.max().clamp(min=1))
    pad = torch.zeros(cap, dtype=local.dtype, device=local.device)
    pad[: local.numel()] = local
    parts = torch.empty(world * cap, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(parts, pad, group=group)
    keep = (torch.arange(cap, device=local.device)[None, :] < sizes[:, None]).reshape(-1)
    return parts[keep]
