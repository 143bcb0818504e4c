"""Tensor-level plumbing over the C ABI (include/sbx.h).

torch owns the device buffers and the stream; every function here only
marshals pointers into libsbx.so.  Nothing in this module computes on the CPU:
if the HIP library or a GPU is missing the calls raise.
"""
import ctypes as C
import threading

import torch

from . import capi

_VT = {torch.int32: capi.V_I32, torch.float32: capi.V_F32, torch.int64: capi.V_I64,
       torch.float64: capi.V_F64}
_handles = {}
_lock = threading.Lock()


def _vt(val):
    if val is None:
        return capi.V_NONE
    if val.dtype not in _VT:
        raise TypeError(f"unsupported value dtype {val.dtype}")
    return _VT[val.dtype]


def _it(t, ids=None):
    """Index tuple of a call: `t` is the call's offset array where it has one (row_ptr / col_ptr), else an id array;
    `ids` an id array of the same call (col, row): int64 offsets over int32 ids are SBX_I32_N64."""
    if t.dtype == torch.int32:
        if ids is not None and ids.dtype != torch.int32:
            raise TypeError("32-bit offsets with 64-bit ids: not a tuple the library takes")
        return capi.SBX_I32
    if t.dtype == torch.int64:
        if ids is not None and ids.dtype == torch.int32:
            return capi.SBX_I32_N64
        return capi.SBX_I64
    raise TypeError(f"index tensors must be int32/int64, got {t.dtype}")


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _check_dev(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise ValueError("sparsebase_amd.ops works on device tensors only (no CPU path)")
        if not t.is_contiguous():
            raise ValueError("tensors must be contiguous")
        dev = t.device if dev is None else dev
        if t.device != dev:
            raise ValueError("all tensors must live on the same device")
    return dev


class Handle:
    """One sbx handle per (device); bound to torch's current stream on every call."""

    def __init__(self, device_index):
        self.lib = capi.load()
        self.h = C.c_void_p()
        rc = self.lib.sbx_create(int(device_index), C.byref(self.h))
        if rc != capi.SBX_OK:
            raise capi.SbxError(rc, "sbx_create failed (HIP hot path has no CPU fallback)")
        self.device_index = device_index

    def bind_stream(self):
        s = torch.cuda.current_stream(self.device_index).cuda_stream
        self.check(self.lib.sbx_set_stream(self.h, C.c_void_p(s)))

    def check(self, rc):
        if rc != capi.SBX_OK:
            raise capi.SbxError(rc, self.lib.sbx_last_error(self.h).decode())

    def reserve(self, nbytes):
        self.check(self.lib.sbx_reserve(self.h, C.c_size_t(nbytes)))


def profile_enable(on=True, device=None):
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    hd = handle_for(dev)
    hd.check(hd.lib.sbx_profile_enable(hd.h, int(bool(on))))


def profile_report(device=None):
    """{kernel group: (total_ms, launches)} accumulated since profile_enable(True)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    hd = handle_for(dev)
    out = {}
    for i in range(hd.lib.sbx_profile_kernel_count()):
        ms, cnt = C.c_double(0), C.c_int64(0)
        hd.check(hd.lib.sbx_profile_query(hd.h, i, C.byref(ms), C.byref(cnt)))
        nb = C.c_int64(0)
        hd.check(hd.lib.sbx_profile_query_bytes(hd.h, i, C.byref(nb)))
        if cnt.value:
            out[hd.lib.sbx_profile_kernel_name(i).decode()] = (ms.value, cnt.value, nb.value)
    return out


def handle_for(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    with _lock:
        if idx not in _handles:
            _handles[idx] = Handle(idx)
    hd = _handles[idx]
    hd.bind_stream()
    return hd


# ----------------------------------------------------------------------------- A1 / A4
def coo_is_sorted(row, col):
    hd = handle_for(_check_dev(row, col))
    out = C.c_int(0)
    hd.check(hd.lib.sbx_coo_is_sorted(hd.h, _it(row), row.numel(), _p(row), _p(col), C.byref(out)))
    return bool(out.value)


def coo_sort_(n, m, row, col, val=None):
    """In place, like the COO constructor (format/coo.cc:110-157)."""
    hd = handle_for(_check_dev(row, col, val))
    hd.check(hd.lib.sbx_coo_sort(hd.h, _it(row), _vt(val), n, m, row.numel(), _p(row), _p(col), _p(val)))


def csr_rows_sorted(row_ptr, col):
    hd = handle_for(_check_dev(row_ptr, col))
    out = C.c_int(0)
    hd.check(hd.lib.sbx_csr_rows_sorted(hd.h, _it(row_ptr, col), row_ptr.numel() - 1, _p(row_ptr), _p(col), C.byref(out)))
    return bool(out.value)


def csr_sort_rows_(n, m, row_ptr, col, val=None):
    """In place, like the CSR constructor (format/csr.cc:99-157)."""
    hd = handle_for(_check_dev(row_ptr, col, val))
    hd.check(hd.lib.sbx_csr_sort_rows(hd.h, _it(row_ptr, col), _vt(val), n, m, col.numel(), _p(row_ptr), _p(col), _p(val)))


# ----------------------------------------------------------------------------- A2 / A3
def coo_to_csr(n, m, row, col, val=None, move=False, rows_sorted=False, out=None, offset_dtype=None):
    """offset_dtype=torch.int64 with int32 ids: the <32-bit ids, 64-bit offsets> tuple (SBX_I32_N64)."""
    hd = handle_for(_check_dev(row, col, val))
    nnz = row.numel()
    if out is None:
        rp = torch.empty(n + 1, dtype=offset_dtype or row.dtype, device=row.device)
        co = None if move else torch.empty_like(col)
        vo = None if (move or val is None) else torch.empty_like(val)
    else:
        rp, co, vo = out
    flags = (capi.FLAG_MOVE if move else 0) | (capi.FLAG_ROWS_SORTED if rows_sorted else 0)
    hd.check(hd.lib.sbx_coo_to_csr(hd.h, _it(rp, row), _vt(val), n, m, nnz, _p(row), _p(col), _p(val), _p(rp), _p(co),
                                   _p(vo), flags))
    return (rp, col, val) if move else (rp, co, vo)


def csr_to_coo(n, m, row_ptr, col, val=None, move=False, out=None):
    hd = handle_for(_check_dev(row_ptr, col, val))
    nnz = col.numel()
    if out is None:
        ro = torch.empty(nnz, dtype=col.dtype if col is not None else row_ptr.dtype, device=row_ptr.device)
        co = None if move else torch.empty_like(col)
        vo = None if (move or val is None) else torch.empty_like(val)
    else:
        ro, co, vo = out
    flags = capi.FLAG_MOVE if move else 0
    hd.check(hd.lib.sbx_csr_to_coo(hd.h, _it(row_ptr, ro), _vt(val), n, m, nnz, _p(row_ptr), _p(col), _p(val), _p(ro),
                                   _p(co), _p(vo), flags))
    return (ro, col, val) if move else (ro, co, vo)


def coo_to_csc(n, m, row, col, val=None, offset_dtype=None):
    """COO -> CSC (col_ptr[m+1], row[nnz], val[nnz]); converter_order_two.cc:21-70."""
    hd = handle_for(_check_dev(row, col, val))
    nnz = col.numel()
    cp = torch.empty(m + 1, dtype=offset_dtype or row.dtype, device=row.device)
    ro = torch.empty_like(row)
    vo = None if val is None else torch.empty_like(val)
    hd.check(hd.lib.sbx_coo_to_csc(hd.h, _it(cp, row), _vt(val), n, m, nnz, _p(row), _p(col), _p(val), _p(cp), _p(ro), _p(vo)))
    return cp, ro, vo


def csr_to_csc(n, m, row_ptr, col, val=None):
    """CSR -> CSC (the transpose when rows are sorted); converter_order_two.cc:120-128."""
    hd = handle_for(_check_dev(row_ptr, col, val))
    nnz = col.numel()
    cp = torch.empty(m + 1, dtype=row_ptr.dtype, device=row_ptr.device)
    ro = torch.empty_like(col)
    vo = None if val is None else torch.empty_like(val)
    hd.check(hd.lib.sbx_csr_to_csc(hd.h, _it(row_ptr, col), _vt(val), n, m, nnz, _p(row_ptr), _p(col), _p(val), _p(cp),
                                   _p(ro), _p(vo)))
    return cp, ro, vo


# ----------------------------------------------------------------------------- Matrix Market ingest
def mtx_parse_coordinate(text, n_rows, n_cols, entries, fields, symmetry=0, zero_index=True, upper_triangle=False,
                         index_dtype=torch.int32, value_dtype=None):
    """text: uint8 device tensor with the bytes after the size line.  Returns (row, col, val) trimmed to nnz."""
    hd = handle_for(_check_dev(text))
    expand = symmetry != 0 and not upper_triangle
    cap = max(1, entries * (2 if expand else 1))
    row = torch.empty(cap, dtype=index_dtype, device=text.device)
    col = torch.empty(cap, dtype=index_dtype, device=text.device)
    val = None if (value_dtype is None or fields != 3) else torch.empty(cap, dtype=value_dtype, device=text.device)
    nnz = C.c_int64(0)
    flags = (1 if zero_index else 0) | (2 if upper_triangle else 0)
    hd.check(hd.lib.sbx_mtx_parse_coordinate(hd.h, _it(row), _vt(val), _p(text), text.numel(), n_rows, n_cols, entries,
                                             fields, symmetry, flags, cap, _p(row), _p(col), _p(val), C.byref(nnz)))
    k = nnz.value
    return row[:k], col[:k], (None if val is None else val[:k])


def text_count_tokens(text):
    hd = handle_for(_check_dev(text))
    out = C.c_int64(0)
    hd.check(hd.lib.sbx_text_count_tokens(hd.h, _p(text), text.numel(), C.byref(out)))
    return out.value


def edge_list_parse(text, weighted=False, remove_duplicates=False, remove_self_edges=False, read_undirected=True,
                    square=False, index_dtype=torch.int32, value_dtype=None):
    """Edge list text (uint8 device tensor) -> (n, m, row, col, val) sorted by (row, col); EdgeListReader::ReadCOO."""
    hd = handle_for(_check_dev(text))
    fields = 3 if weighted else 2
    tokens = text_count_tokens(text)
    if tokens % fields:
        raise ValueError(f"edge list holds {tokens} tokens, not a multiple of {fields}")
    entries = tokens // fields
    cap = max(1, entries * (2 if read_undirected else 1))
    row = torch.empty(cap, dtype=index_dtype, device=text.device)
    col = torch.empty(cap, dtype=index_dtype, device=text.device)
    val = None if (value_dtype is None or not weighted) else torch.empty(cap, dtype=value_dtype, device=text.device)
    dims = (C.c_int64 * 3)()
    flags = (1 if remove_duplicates else 0) | (2 if remove_self_edges else 0) | (4 if read_undirected else 0) | \
        (8 if square else 0)
    hd.check(hd.lib.sbx_edge_list_parse(hd.h, _it(row), _vt(val), _p(text), text.numel(), entries, int(weighted), flags,
                                        cap, _p(row), _p(col), _p(val), dims))
    k = dims[2]
    return dims[0], dims[1], row[:k], col[:k], (None if val is None else val[:k])


# ----------------------------------------------------------------------------- features
def csr_degrees(row_ptr, id_dtype=None):
    """id_dtype=torch.int32 over an int64 row_ptr: the <32-bit ids, 64-bit offsets> tuple (degrees are ids)."""
    hd = handle_for(_check_dev(row_ptr))
    n = row_ptr.numel() - 1
    out = torch.empty(n, dtype=id_dtype or row_ptr.dtype, device=row_ptr.device)
    hd.check(hd.lib.sbx_csr_degrees(hd.h, _it(row_ptr, out), n, _p(row_ptr), _p(out)))
    return out


def csr_degree_distribution(row_ptr, nnz, dtype=torch.float32):
    hd = handle_for(_check_dev(row_ptr))
    n = row_ptr.numel() - 1
    out = torch.empty(n, dtype=dtype, device=row_ptr.device)
    hd.check(hd.lib.sbx_csr_degree_distribution(hd.h, _it(row_ptr), n, nnz, _p(row_ptr), out.element_size(), _p(out)))
    return out


def csr_bandwidth(row_ptr, col):
    hd = handle_for(_check_dev(row_ptr, col))
    out = C.c_int64(0)
    hd.check(hd.lib.sbx_csr_bandwidth(hd.h, _it(row_ptr, col), row_ptr.numel() - 1, col.numel(), _p(row_ptr), _p(col),
                                      C.byref(out)))
    return out.value


def csr_profile(row_ptr, col):
    hd = handle_for(_check_dev(row_ptr, col))
    out = C.c_int64(0)
    hd.check(hd.lib.sbx_csr_profile(hd.h, _it(row_ptr, col), row_ptr.numel() - 1, col.numel(), _p(row_ptr), _p(col),
                                    C.byref(out)))
    return out.value


# ----------------------------------------------------------------------------- reorderers
def degree_reorder(row_ptr, ascending=True, out=None, id_dtype=None):
    hd = handle_for(_check_dev(row_ptr))
    n = row_ptr.numel() - 1
    inv = torch.empty(n, dtype=id_dtype or row_ptr.dtype, device=row_ptr.device) if out is None else out
    hd.check(hd.lib.sbx_degree_reorder(hd.h, _it(row_ptr, inv), n, _p(row_ptr), int(bool(ascending)), _p(inv)))
    return inv


def rcm_reorder(row_ptr, col, out=None, return_stats=False):
    hd = handle_for(_check_dev(row_ptr, col))
    n = row_ptr.numel() - 1
    inv = torch.empty(n, dtype=col.dtype, device=row_ptr.device) if out is None else out
    stats = capi.RcmStats()
    hd.check(hd.lib.sbx_rcm_reorder(hd.h, _it(row_ptr, col), n, col.numel(), _p(row_ptr), _p(col), _p(inv),
                                    C.byref(stats)))
    if return_stats:
        return inv, {k: getattr(stats, k) for k, _ in capi.RcmStats._fields_}
    return inv


def gray_row_keys(m, row_ptr, col, resolution, nnz_threshold):
    hd = handle_for(_check_dev(row_ptr, col))
    n = row_ptr.numel() - 1
    deg = torch.empty(n, dtype=col.dtype, device=row_ptr.device)
    key = torch.empty(n, dtype=torch.int64, device=row_ptr.device)  # uint64 bit pattern
    counts = (C.c_int64 * 4)()
    hd.check(hd.lib.sbx_gray_row_keys(hd.h, _it(row_ptr, col), n, m, col.numel(), _p(row_ptr), _p(col), int(resolution),
                                      int(nnz_threshold), _p(deg), _p(key), counts))
    return deg, key, list(counts)


def gray_reorder(m, row_ptr, col, resolution, nnz_threshold, group_size, exact_ties=False):
    """GrayReorder with the ordering stage on the device (stable ties; exact_ties=True is refused by the library: the
    exact mode is the host layer's reorder::GrayReorder)."""
    hd = handle_for(_check_dev(row_ptr, col))
    n = row_ptr.numel() - 1
    inv = torch.empty(n, dtype=col.dtype, device=row_ptr.device)
    hd.check(hd.lib.sbx_gray_reorder(hd.h, _it(row_ptr, col), n, m, col.numel(), _p(row_ptr), _p(col), int(resolution),
                                     int(nnz_threshold), int(group_size), 1 if exact_ties else 0, _p(inv)))
    return inv


# ----------------------------------------------------------------------------- permutation
def inverse_permutation(perm):
    hd = handle_for(_check_dev(perm))
    inv = torch.empty_like(perm)
    hd.check(hd.lib.sbx_inverse_permutation(hd.h, _it(perm), perm.numel(), _p(perm), _p(inv)))
    return inv


def permute_csr(n, m, row_ptr, col, val, row_order, col_order, out=None):
    hd = handle_for(_check_dev(row_ptr, col, val, row_order, col_order))
    if out is None:
        rpo = torch.empty_like(row_ptr)
        co = torch.empty_like(col)
        vo = None if val is None else torch.empty_like(val)
    else:
        rpo, co, vo = out
    hd.check(hd.lib.sbx_permute_csr(hd.h, _it(row_ptr, col), _vt(val), n, m, col.numel(), _p(row_ptr), _p(col), _p(val),
                                    _p(row_order), _p(col_order), _p(rpo), _p(co), _p(vo)))
    return rpo, co, vo


def permute_csr_rows_nnz(n, row_ptr, row_order, row_begin, row_end):
    """Entries of the new rows [row_begin, row_end): the size of that shard's col / val slab."""
    hd = handle_for(_check_dev(row_ptr, row_order))
    got = C.c_int64(0)
    hd.check(hd.lib.sbx_permute_csr_rows_nnz(hd.h, _it(row_ptr, row_order), n, _p(row_ptr), _p(row_order), row_begin, row_end,
                                             C.byref(got)))
    return got.value


def permute_csr_rows(n, m, row_ptr, col, val, row_order, col_order, row_begin, row_end, capacity=None):
    """One row-range shard of the permuted matrix (the multi-GPU decomposition)."""
    hd = handle_for(_check_dev(row_ptr, col, val, row_order, col_order))
    nr = row_end - row_begin
    if capacity is None:  # exactly the shard's entries, not the whole matrix's
        capacity = permute_csr_rows_nnz(n, row_ptr, row_order, row_begin, row_end)
    rpo = torch.empty(nr + 1, dtype=row_ptr.dtype, device=row_ptr.device)
    co = torch.empty(max(capacity, 1), dtype=col.dtype, device=col.device)  # (an empty slab still needs an address)
    vo = None if val is None else torch.empty(max(capacity, 1), dtype=val.dtype, device=val.device)
    got = C.c_int64(0)
    hd.check(hd.lib.sbx_permute_csr_rows(hd.h, _it(row_ptr, col), _vt(val), n, m, col.numel(), _p(row_ptr), _p(col),
                                         _p(val), _p(row_order), _p(col_order), row_begin, row_end, _p(rpo), _p(co),
                                         _p(vo), capacity, C.byref(got)))
    k = got.value
    return rpo, co[:k], (None if vo is None else vo[:k])


def permute_array(order, vals):
    hd = handle_for(_check_dev(order, vals))
    out = torch.empty_like(vals)
    hd.check(hd.lib.sbx_permute_array(hd.h, _it(order), _vt(vals), order.numel(), _p(order), _p(vals), _p(out)))
    return out


# ----------------------------------------------------------------------------- sharded steps
class Comm:
    """An sbx communicator: RCCL (`Comm.rccl`, one GPU per rank, the id travels over torch.distributed) or an
    all-gather hook over a torch.distributed group of any backend (`Comm.hook`, e.g. gloo with ranks sharing a GPU)."""

    def __init__(self, lib, handle, keep=None):
        self.lib, self.c, self._keep = lib, handle, keep

    @classmethod
    def rccl(cls, device_index, group=None):
        import torch.distributed as dist
        lib = capi.load()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [None]
        if rank == 0:
            buf = C.create_string_buffer(capi.COMM_ID_BYTES)
            rc = lib.sbx_comm_unique_id(buf)
            if rc != capi.SBX_OK:
                raise capi.SbxError(rc, "sbx_comm_unique_id failed (librccl not loadable?)")
            box[0] = buf.raw
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        c = C.c_void_p()
        rc = lib.sbx_comm_create_rccl(int(device_index), rank, world, C.c_char_p(box[0]), C.byref(c))
        if rc != capi.SBX_OK:
            raise capi.SbxError(rc, "sbx_comm_create_rccl failed")
        return cls(lib, c)

    @classmethod
    def hook(cls, device_index, group=None):
        """All-gather through torch.distributed on host buffers (staged with sbx_memcpy_*): for backends without
        device collectives and for ranks that share one GPU."""
        import numpy as np
        import torch.distributed as dist
        lib = capi.load()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        hd = handle_for(torch.device("cuda", device_index))

        def allgather(user, send, recv, nbytes, stream):
            try:
                mine = np.empty(nbytes, np.uint8)
                torch.cuda.current_stream(device_index).synchronize()
                if lib.sbx_memcpy_d2h(hd.h, mine.ctypes.data_as(C.c_void_p), C.c_void_p(send), nbytes) != capi.SBX_OK:
                    return 3
                parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(parts, torch.from_numpy(mine), group=group)
                whole = torch.cat(parts).numpy()
                if lib.sbx_memcpy_h2d(hd.h, C.c_void_p(recv), whole.ctypes.data_as(C.c_void_p), nbytes * world) != capi.SBX_OK:
                    return 3
                return 0
            except Exception:  # noqa: BLE001  (must not unwind through the C frame)
                return 6

        cb = capi.ALLGATHER_FN(allgather)
        c = C.c_void_p()
        rc = lib.sbx_comm_create(rank, world, cb, None, C.byref(c))
        if rc != capi.SBX_OK:
            raise capi.SbxError(rc, "sbx_comm_create failed")
        return cls(lib, c, keep=cb)

    def close(self):
        if self.c:
            self.lib.sbx_comm_destroy(self.c)
            self.c = None


def _splits(ranges):
    if ranges is None:
        return None
    cuts = [ranges[0][0]] + [hi for _, hi in ranges]
    return (C.c_int64 * len(cuts))(*cuts)


def permute_csr_sharded(comm, n, m, row_ptr, col, val, row_order, col_order, ranges=None, out=None, capacity=None):
    """sbx_permute_csr_sharded: returns (global row_ptr, this rank's col, val slabs, shard offsets list).
    `out` = (row_ptr_out (n + 1), col_out, val_out) pre-allocated slabs (capacity = col_out.numel())."""
    hd = handle_for(_check_dev(row_ptr, col, val, row_order, col_order))
    rank, world = C.c_int(0), C.c_int(0)
    hd.lib.sbx_comm_rank(comm.c, C.byref(rank), C.byref(world))
    if out is None:
        if capacity is None:
            lo, hi = ranges[rank.value] if ranges is not None else _equal_range(n, world.value, rank.value)
            capacity = permute_csr_rows_nnz(n, row_ptr, row_order, lo, hi)
        rpo = torch.empty(n + 1, dtype=row_ptr.dtype, device=row_ptr.device)
        co = torch.empty(max(capacity, 1), dtype=col.dtype, device=col.device)
        vo = None if val is None else torch.empty(max(capacity, 1), dtype=val.dtype, device=val.device)
    else:
        rpo, co, vo = out
        capacity = co.numel()
    offs = (C.c_int64 * (world.value + 1))()
    hd.check(hd.lib.sbx_permute_csr_sharded(hd.h, comm.c, _it(row_ptr), _vt(val), n, m, col.numel(), _p(row_ptr), _p(col),
                                            _p(val), _p(row_order), _p(col_order), _splits(ranges), _p(rpo), _p(co),
                                            _p(vo), capacity, offs))
    offs = list(offs)
    k = offs[rank.value + 1] - offs[rank.value]
    return rpo, co[:k], (None if vo is None else vo[:k]), offs


def coo_to_csr_sharded(comm, n, m, row, col, val, ranges=None, out=None, capacity=None):
    """sbx_coo_to_csr_sharded on a replicated row-sorted COO."""
    hd = handle_for(_check_dev(row, col, val))
    rank, world = C.c_int(0), C.c_int(0)
    hd.lib.sbx_comm_rank(comm.c, C.byref(rank), C.byref(world))
    if out is None:
        if capacity is None:
            lo, hi = ranges[rank.value] if ranges is not None else _equal_range(n, world.value, rank.value)
            b = torch.searchsorted(row, torch.tensor([lo, hi], dtype=row.dtype, device=row.device), right=False)
            capacity = int(b[1] - b[0])
        rpo = torch.empty(n + 1, dtype=row.dtype, device=row.device)
        co = torch.empty(max(capacity, 1), dtype=col.dtype, device=col.device)
        vo = None if val is None else torch.empty(max(capacity, 1), dtype=val.dtype, device=val.device)
    else:
        rpo, co, vo = out
        capacity = co.numel()
    offs = (C.c_int64 * (world.value + 1))()
    hd.check(hd.lib.sbx_coo_to_csr_sharded(hd.h, comm.c, _it(row), _vt(val), n, m, row.numel(), _p(row), _p(col), _p(val),
                                           _splits(ranges), _p(rpo), _p(co), _p(vo), capacity, offs))
    offs = list(offs)
    k = offs[rank.value + 1] - offs[rank.value]
    return rpo, co[:k], (None if vo is None else vo[:k]), offs


def csr_to_coo_sharded(comm, n, m, row_ptr, col, val, ranges=None, out=None, capacity=None):
    """sbx_csr_to_coo_sharded on a replicated CSR: this rank's slab of the COO (global row ids).
    Returns (row, col, val slabs, shard offsets list)."""
    hd = handle_for(_check_dev(row_ptr, col, val))
    rank, world = C.c_int(0), C.c_int(0)
    hd.lib.sbx_comm_rank(comm.c, C.byref(rank), C.byref(world))
    if out is None:
        if capacity is None:
            lo, hi = ranges[rank.value] if ranges is not None else _equal_range(n, world.value, rank.value)
            capacity = int(row_ptr[hi] - row_ptr[lo])
        ro = torch.empty(max(capacity, 1), dtype=row_ptr.dtype, device=row_ptr.device)
        co = torch.empty(max(capacity, 1), dtype=col.dtype, device=col.device)
        vo = None if val is None else torch.empty(max(capacity, 1), dtype=val.dtype, device=val.device)
    else:
        ro, co, vo = out
        capacity = co.numel()
    offs = (C.c_int64 * (world.value + 1))()
    hd.check(hd.lib.sbx_csr_to_coo_sharded(hd.h, comm.c, _it(row_ptr), _vt(val), n, m, col.numel(), _p(row_ptr), _p(col),
                                           _p(val), _splits(ranges), _p(ro), _p(co), _p(vo), capacity, offs))
    offs = list(offs)
    k = offs[rank.value + 1] - offs[rank.value]
    return ro[:k], co[:k], (None if vo is None else vo[:k]), offs


def balanced_row_splits(n, row_ptr, row_order, world):
    """sbx_balanced_row_splits: new-row ranges of (nearly) equal entry counts, as a list of (lo, hi)."""
    hd = handle_for(_check_dev(row_ptr, row_order))
    cuts = (C.c_int64 * (world + 1))()
    hd.check(hd.lib.sbx_balanced_row_splits(hd.h, _it(row_ptr), n, _p(row_ptr), _p(row_order), world, cuts))
    cuts = list(cuts)
    return list(zip(cuts[:-1], cuts[1:]))


def _equal_range(n, world, rank):
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
