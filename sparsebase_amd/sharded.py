"""Row-range sharded permutation apply and format conversion across the GPUs of one node
(SURVEY.md §8e).

Decomposition: new row i depends only on old row old_of_new[i] and the replicated
col_order table, so rank r owns new rows [lo_r, hi_r).  Every rank holds the input
CSR and the order vectors (replicated); it produces its slab of col/val locally
(sbx_permute_csr_rows) and the only exchange is

  1. all_gather of the per-rank nnz totals      (world_size int64 values)
  2. all_gather of the rebased row_ptr segments (Z*n bytes in total)

after which every rank holds the complete permuted row_ptr; col/val stay
row-sharded (a distributed CSR) unless gather_entries=True.  The collective is
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests that exercise exactly this stitching code with a test double for the
shard computation).
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


def balanced_row_ranges(new_row_lengths_prefix, world_size):
    """Ranges balanced by nnz instead of rows (power-law inputs): split points are
    found on the host from the scanned new-row lengths (a length n+1 CPU tensor)."""
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


def stitch_row_ptr(local_row_ptr, ranges, group=None):
    """All-gather the rebased per-shard row_ptr segments into the global row_ptr.

    local_row_ptr: (hi-lo+1,) tensor, rebased to 0, on this rank's device.
    Returns (global_row_ptr (n+1,), shard_offsets (world_size+1,) int64 tensor).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = ranges[rank]
    assert local_row_ptr.numel() == hi - lo + 1
    dev, dt = local_row_ptr.device, local_row_ptr.dtype
    # (1) shard nnz totals -> global offsets
    mine = local_row_ptr[-1:].to(torch.int64)
    totals = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(totals, mine, group=group)
    totals = torch.cat(totals)
    offsets = torch.zeros(world + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(totals, 0)
    # (2) row_ptr segments, padded to equal length (all-gather-v emulation)
    max_rows = max(h - l for l, h in ranges)
    seg = torch.zeros(max_rows, dtype=dt, device=dev)
    seg[: hi - lo] = local_row_ptr[:-1] + offsets[rank].to(dt)
    gathered = [torch.empty(max_rows, dtype=dt, device=dev) for _ in range(world)]
    dist.all_gather(gathered, seg, group=group)
    n = ranges[-1][1]
    out = torch.empty(n + 1, dtype=dt, device=dev)
    for r, (l, h) in enumerate(ranges):
        out[l:h] = gathered[r][: h - l]
    out[n] = offsets[world].to(dt)
    return out, offsets


def permute_csr_sharded(n, m, row_ptr, col, val, row_order, col_order, group=None, ranges=None,
                        shard_fn=None, gather_entries=False):
    """Sharded Permute2D.  Returns (global_row_ptr, local_col, local_val, (lo, hi), offsets).

    shard_fn(lo, hi) -> (rebased_row_ptr, col, val) defaults to the HIP path
    (ops.permute_csr_rows); tests inject a CPU double to exercise the collectives.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ranges = row_ranges(n, world) if ranges is None else ranges
    lo, hi = ranges[rank]
    if shard_fn is None:
        from . import ops

        def shard_fn(a, b):
            return ops.permute_csr_rows(n, m, row_ptr, col, val, row_order, col_order, a, b)
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


def coo_to_csr_sharded(n, m, row, col, val, group=None, ranges=None, shard_fn=None, gather_entries=False):
    """Sharded COO -> CSR of a row-sorted COO (replicated input, like the permute): rank r owns rows
    [lo_r, hi_r); its nonzeros are the contiguous slice found by binary search on the sorted row array,
    converted locally (sbx_coo_to_csr on rebased row ids), and the row_ptr segments are stitched with
    the same two all-gathers.  Returns (global_row_ptr, local_col, local_val, (lo, hi), offsets).

    shard_fn(lo, hi, a, b) -> (rebased_row_ptr, col, val) for rows [lo, hi) = nonzeros [a, b)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ranges = row_ranges(n, world) if ranges is None else ranges
    lo, hi = ranges[rank]
    bounds = torch.searchsorted(row, torch.tensor([lo, hi], dtype=row.dtype, device=row.device), right=False)
    a, b = int(bounds[0]), int(bounds[1])
    if shard_fn is None:
        def shard_fn(lo_, hi_, a_, b_):
            return coo_to_csr_shard(m, row, col, val, lo_, hi_, a_, b_)
    lrp, lcol, lval = shard_fn(lo, hi, a, b)
    grp, offsets = stitch_row_ptr(lrp, ranges, group)
    if gather_entries:
        lcol = _gather_v(lcol, offsets, group)
        lval = None if lval is None else _gather_v(lval, offsets, group)
    return grp, lcol, lval, (lo, hi), offsets


def csr_to_coo_sharded(n, m, row_ptr, col, val, group=None, ranges=None, shard_fn=None, gather_entries=False):
    """Sharded CSR -> COO: rank r expands the row ids of rows [lo_r, hi_r) (nonzeros
    [row_ptr[lo], row_ptr[hi])); no collective is needed unless the caller wants every rank to hold the
    whole COO (gather_entries).  Returns (local_row, local_col, local_val, (lo, hi), (a, b))."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ranges = row_ranges(n, world) if ranges is None else ranges
    lo, hi = ranges[rank]
    a, b = int(row_ptr[lo]), int(row_ptr[hi])
    if shard_fn is None:
        def shard_fn(lo_, hi_, a_, b_):
            return csr_to_coo_shard(m, row_ptr, col, val, lo_, hi_, a_, b_)
    lrow, lcol, lval = shard_fn(lo, hi, a, b)
    if gather_entries:
        cuts = torch.tensor([int(row_ptr[l]) for l, _ in ranges] + [int(row_ptr[ranges[-1][1]])], dtype=torch.int64,
                            device=row_ptr.device)
        lrow = _gather_v(lrow, cuts, group)
        lcol = _gather_v(lcol, cuts, group)
        lval = None if lval is None else _gather_v(lval, cuts, group)
    return lrow, lcol, lval, (lo, hi), (a, b)


def _gather_v(local, offsets, group=None):
    world = dist.get_world_size(group)
    sizes = (offsets[1:] - offsets[:-1]).tolist()
    cap = int(max(sizes)) if sizes else 0
    pad = torch.zeros(cap, dtype=local.dtype, device=local.device)
    pad[: local.numel()] = local
    parts = [torch.empty(cap, dtype=local.dtype, device=local.device) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: int(s)] for p, s in zip(parts, sizes)])
