"""ctypes binding of the C ABI in include/sbx.h (sparsebase_amd/lib/libsbx.so).

There is deliberately NO fallback: if the library is missing or no GPU is
usable, loading / handle creation raises.  torch is used only as the owner of
device memory and streams; every pointer handed to the ABI is a raw device
pointer.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsbx.so")

SBX_OK = 0
SBX_I32, SBX_I64, SBX_I32_N64 = 0, 1, 2
V_NONE, V_I32, V_U32, V_F32, V_I64, V_U64, V_F64 = range(7)
FLAG_MOVE, FLAG_ROWS_SORTED = 1, 2

_STATUS = {0: "ok", 1: "bad argument", 2: "no usable HIP device", 3: "HIP runtime error",
           4: "out of device memory", 5: "unsupported type tuple or shape", 6: "internal error"}


class SbxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"sbx status {status} ({_STATUS.get(status, '?')}): {msg}")
        self.status = status


class RcmStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("components", "isolated", "small_components", "large_components",
                                         "bfs_sweeps", "bfs_levels", "edges_scanned", "edges_scanned_bottom_up",
                                         "largest_component", "reference_sweeps", "unordered_sweeps")]


# every symbol include/sbx.h declares (tests/test_abi.py checks header <-> library <-> this table)
_i64, _int, _u, _vp, _sz = C.c_int64, C.c_int, C.c_uint, C.c_void_p, C.c_size_t
_H = C.c_void_p
COMM_ID_BYTES = 128
# int (*sbx_allgather_fn)(void *user, const void *send_dev, void *recv_dev, size_t bytes, void *stream)
ALLGATHER_FN = C.CFUNCTYPE(_int, _vp, _vp, _vp, _sz, _vp)
# int (*sbx_oom_hook)(void *user, size_t bytes_wanted)
OOM_HOOK_FN = C.CFUNCTYPE(_int, _vp, _sz)
PROTOTYPES = {
    "sbx_version": ([], _int),
    "sbx_status_string": ([_int], C.c_char_p),
    "sbx_device_count": ([C.POINTER(_int)], _int),
    "sbx_can_access_peer": ([_int, _int, C.POINTER(_int)], _int),
    "sbx_create": ([_int, C.POINTER(_H)], _int),
    "sbx_destroy": ([_H], _int),
    "sbx_set_stream": ([_H, _vp], _int),
    "sbx_get_device": ([_H, C.POINTER(_int)], _int),
    "sbx_reserve": ([_H, _sz], _int),
    "sbx_sync": ([_H], _int),
    "sbx_last_error": ([_H], C.c_char_p),
    "sbx_set_oom_hook": ([_H, OOM_HOOK_FN, _vp], _int),
    "sbx_profile_enable": ([_H, _int], _int),
    "sbx_profile_kernel_count": ([], _int),
    "sbx_profile_kernel_name": ([_int], C.c_char_p),
    "sbx_profile_query": ([_H, _int, C.POINTER(C.c_double), C.POINTER(_i64)], _int),
    "sbx_profile_query_bytes": ([_H, _int, C.POINTER(_i64)], _int),
    "sbx_malloc": ([_H, _sz, C.POINTER(_vp)], _int),
    "sbx_free": ([_H, _vp], _int),
    "sbx_host_alloc": ([_H, _sz, C.POINTER(_vp)], _int),
    "sbx_host_free": ([_H, _vp], _int),
    "sbx_memcpy_h2d": ([_H, _vp, _vp, _sz], _int),
    "sbx_memcpy_d2h": ([_H, _vp, _vp, _sz], _int),
    "sbx_memcpy_d2d": ([_H, _vp, _vp, _sz], _int),
    "sbx_memcpy_peer": ([_H, _vp, _int, _vp, _int, _sz], _int),
    "sbx_coo_is_sorted": ([_H, _int, _i64, _vp, _vp, C.POINTER(_int)], _int),
    "sbx_coo_sort": ([_H, _int, _int, _i64, _i64, _i64, _vp, _vp, _vp], _int),
    "sbx_csr_rows_sorted": ([_H, _int, _i64, _vp, _vp, C.POINTER(_int)], _int),
    "sbx_csr_sort_rows": ([_H, _int, _int, _i64, _i64, _i64, _vp, _vp, _vp], _int),
    "sbx_coo_to_csr": ([_H, _int, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _u], _int),
    "sbx_csr_to_coo": ([_H, _int, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _u], _int),
    "sbx_coo_to_csc": ([_H, _int, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp], _int),
    "sbx_csr_to_csc": ([_H, _int, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp], _int),
    "sbx_mtx_parse_coordinate": ([_H, _int, _int, _vp, _i64, _i64, _i64, _i64, _int, _int, _u, _i64, _vp, _vp, _vp,
                                 C.POINTER(_i64)], _int),
    "sbx_text_count_tokens": ([_H, _vp, _i64, C.POINTER(_i64)], _int),
    "sbx_edge_list_parse": ([_H, _int, _int, _vp, _i64, _i64, _int, _u, _i64, _vp, _vp, _vp, C.POINTER(_i64)], _int),
    "sbx_csr_degrees": ([_H, _int, _i64, _vp, _vp], _int),
    "sbx_csr_degree_distribution": ([_H, _int, _i64, _i64, _vp, _int, _vp], _int),
    "sbx_csr_bandwidth": ([_H, _int, _i64, _i64, _vp, _vp, C.POINTER(_i64)], _int),
    "sbx_csr_profile": ([_H, _int, _i64, _i64, _vp, _vp, C.POINTER(_i64)], _int),
    "sbx_degree_reorder": ([_H, _int, _i64, _vp, _int, _vp], _int),
    "sbx_rcm_reorder": ([_H, _int, _i64, _i64, _vp, _vp, _vp, C.POINTER(RcmStats)], _int),
    "sbx_gray_row_keys": ([_H, _int, _i64, _i64, _i64, _vp, _vp, _int, _int, _vp, _vp, C.POINTER(_i64)], _int),
    "sbx_gray_reorder": ([_H, _int, _i64, _i64, _i64, _vp, _vp, _int, _int, _int, _int, _vp], _int),
    "sbx_inverse_permutation": ([_H, _int, _i64, _vp, _vp], _int),
    "sbx_permute_csr": ([_H, _int, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _int),
    "sbx_permute_csr_rows": ([_H, _int, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp,
                              _vp, _i64, C.POINTER(_i64)], _int),
    "sbx_permute_array": ([_H, _int, _int, _i64, _vp, _vp, _vp], _int),
    "sbx_comm_create": ([_int, _int, ALLGATHER_FN, _vp, C.POINTER(_vp)], _int),
    "sbx_comm_unique_id": ([_vp], _int),
    "sbx_comm_create_rccl": ([_int, _int, _int, _vp, C.POINTER(_vp)], _int),
    "sbx_comm_rank": ([_vp, C.POINTER(_int), C.POINTER(_int)], _int),
    "sbx_comm_destroy": ([_vp], _int),
    "sbx_permute_csr_rows_nnz": ([_H, _int, _i64, _vp, _vp, _i64, _i64, C.POINTER(_i64)], _int),
    "sbx_permute_csr_sharded": ([_H, _vp, _int, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i64), _vp,
                                 _vp, _vp, _i64, C.POINTER(_i64)], _int),
    "sbx_coo_to_csr_sharded": ([_H, _vp, _int, _int, _i64, _i64, _i64, _vp, _vp, _vp, C.POINTER(_i64), _vp, _vp, _vp,
                                _i64, C.POINTER(_i64)], _int),
    "sbx_csr_to_coo_sharded": ([_H, _vp, _int, _int, _i64, _i64, _i64, _vp, _vp, _vp, C.POINTER(_i64), _vp, _vp, _vp,
                                _i64, C.POINTER(_i64)], _int),
    "sbx_balanced_row_splits": ([_H, _int, _i64, _vp, _vp, _int, C.POINTER(_i64)], _int),
}

_lib = None


def load():
    """Load libsbx.so (building is __graft_entry__.build()'s job, not an import side effect)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -m sparsebase_amd.build` "
                              "(there is no CPU fallback for the HIP hot path)")
        lib = C.CDLL(LIB_PATH)
        for name, (argtypes, restype) in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError here == header/library drift: fail loudly
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = lib
    return _lib
