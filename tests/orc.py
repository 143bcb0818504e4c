"""ctypes front-ends for the two CPU checkers (TEST INFRASTRUCTURE ONLY).

* ``Oracle`` — oracle/libsbx_oracle.so, our restatement (oracle/sbx_oracle.cc).
* ``Ref``    — oracle/_ref/libsbref.so, the real SparseBase reference compiled by
  oracle/Makefile (present in the build container and, prebuilt, on the GPU box).

Both expose the same numpy-in / numpy-out methods so tests can swap them.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

V_NONE, V_I32, V_U32, V_F32, V_I64, V_U64, V_F64 = range(7)
_VT_OF_DTYPE = {
    np.dtype(np.int32): V_I32, np.dtype(np.uint32): V_U32, np.dtype(np.float32): V_F32,
    np.dtype(np.int64): V_I64, np.dtype(np.uint64): V_U64, np.dtype(np.float64): V_F64,
}


def vt_of(val):
    return V_NONE if val is None else _VT_OF_DTYPE[val.dtype]


def it_of(arr):
    if arr.dtype in (np.dtype(np.int32), np.dtype(np.uint32)):
        return 0
    if arr.dtype == np.dtype(np.int64):
        return 1
    raise TypeError(arr.dtype)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
    return os.path.join(ORACLE_DIR, "libsbx_oracle.so")


class Oracle:
    def __init__(self):
        path = os.environ.get("SBX_ORACLE_LIB") or os.path.join(ORACLE_DIR, "libsbx_oracle.so")  # (make -C oracle asan)
        src = os.path.join(ORACLE_DIR, "sbx_oracle.cc")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build_oracle()
        self.lib = C.CDLL(path)
        self.lib.orc_coo_is_sorted.restype = C.c_int
        self.lib.orc_csr_rows_sorted.restype = C.c_int
        self.lib.orc_gray_reorder.restype = C.c_int
        self.lib.orc_gray_row_keys.restype = C.c_int
        self.lib.orc_csr_bandwidth.restype = C.c_int64
        self.lib.orc_csr_profile.restype = C.c_int64

    def coo_is_sorted(self, row, col):
        return bool(self.lib.orc_coo_is_sorted(it_of(row), C.c_int64(len(row)), _p(row), _p(col)))

    def coo_sort(self, row, col, val=None, n=None, m=None):
        row, col = row.copy(), col.copy()
        val = None if val is None else val.copy()
        self.lib.orc_coo_sort(it_of(row), vt_of(val), C.c_int64(len(row)), _p(row), _p(col), _p(val))
        return row, col, val

    def csr_rows_sorted(self, rp, col):
        return bool(self.lib.orc_csr_rows_sorted(it_of(rp), C.c_int64(len(rp) - 1), _p(rp), _p(col)))

    def csr_sort_rows(self, rp, col, val=None, m=None):
        col = col.copy()
        val = None if val is None else val.copy()
        self.lib.orc_csr_sort_rows(it_of(rp), vt_of(val), C.c_int64(len(rp) - 1), _p(rp), _p(col), _p(val))
        return col, val

    def coo_to_csr(self, n, row, col, val=None, m=None):
        nnz = len(row)
        rp = np.empty(n + 1, row.dtype)
        co = np.empty(nnz, row.dtype)
        vo = None if val is None else np.empty_like(val)
        self.lib.orc_coo_to_csr(it_of(row), vt_of(val), C.c_int64(n), C.c_int64(nnz), _p(row), _p(col),
                                _p(val), _p(rp), _p(co), _p(vo))
        return rp, co, vo

    def csr_to_coo(self, rp, col, val=None, m=None):
        n, nnz = len(rp) - 1, len(col)
        ro = np.empty(nnz, rp.dtype)
        co = np.empty(nnz, rp.dtype)
        vo = None if val is None else np.empty_like(val)
        self.lib.orc_csr_to_coo(it_of(rp), vt_of(val), C.c_int64(n), C.c_int64(nnz), _p(rp), _p(col),
                                _p(val), _p(ro), _p(co), _p(vo))
        return ro, co, vo

    def mtx_parse(self, text, entries, fields, symmetry=0, zero_index=True, upper=False, index_dtype=np.int32,
                  value_dtype=None):
        """Coordinate section (bytes after the size line) -> (row, col, val) in file order; raises on a failed stream."""
        cap = max(1, entries * 2)
        row, col = np.empty(cap, index_dtype), np.empty(cap, index_dtype)
        val = None if value_dtype is None else np.empty(cap, value_dtype)
        nnz = C.c_int64(0)
        buf = bytes(text)
        rc = self.lib.orc_mtx_parse(0 if index_dtype == np.int32 else 1, vt_of(val), buf, C.c_int64(len(buf)),
                                    C.c_int64(entries), int(fields), int(symmetry), int(zero_index), int(upper),
                                    _p(row), _p(col), _p(val), C.byref(nnz))
        if rc != 0:
            raise ValueError("mtx_parse: stream failed")
        k = nnz.value
        return row[:k].copy(), col[:k].copy(), (None if val is None else val[:k].copy())

    def edge_list_parse(self, text, weighted=False, remove_duplicates=False, remove_self_edges=False,
                        read_undirected=True, square=False, index_dtype=np.int32, value_dtype=None):
        buf = bytes(text)
        cap = max(1, 2 * (buf.count(b"\n") + 2) * 2)
        row, col = np.empty(cap, index_dtype), np.empty(cap, index_dtype)
        val = None if value_dtype is None else np.empty(cap, value_dtype)
        dims = (C.c_int64 * 3)()
        self.lib.orc_edge_list_parse(0 if index_dtype == np.int32 else 1, vt_of(val), buf, C.c_int64(len(buf)),
                                     int(weighted), int(remove_duplicates), int(remove_self_edges), int(read_undirected),
                                     int(square), _p(row), _p(col), _p(val), dims)
        k = dims[2]
        return dims[0], dims[1], row[:k].copy(), col[:k].copy(), (None if val is None else val[:k].copy())

    def csr_bandwidth(self, rp, col):
        return int(self.lib.orc_csr_bandwidth(it_of(rp), C.c_int64(len(rp) - 1), _p(rp), _p(col)))

    def csr_profile(self, rp, col):
        return int(self.lib.orc_csr_profile(it_of(rp), C.c_int64(len(rp) - 1), _p(rp), _p(col)))

    def csr_degrees(self, rp):
        out = np.empty(len(rp) - 1, rp.dtype)
        self.lib.orc_csr_degrees(it_of(rp), C.c_int64(len(rp) - 1), _p(rp), _p(out))
        return out

    def csr_degree_distribution(self, rp, nnz, dtype=np.float32):
        out = np.empty(len(rp) - 1, dtype)
        self.lib.orc_csr_degree_distribution(it_of(rp), out.itemsize, C.c_int64(len(rp) - 1), C.c_int64(nnz), _p(rp),
                                             _p(out))
        return out

    def coo_to_csc(self, n, m, row, col, val=None):
        nnz = len(row)
        cp = np.empty(m + 1, row.dtype)
        ro = np.empty(nnz, row.dtype)
        vo = None if val is None else np.empty_like(val)
        self.lib.orc_coo_to_csc(it_of(row), vt_of(val), C.c_int64(n), C.c_int64(m), C.c_int64(nnz), _p(row), _p(col),
                                _p(val), _p(cp), _p(ro), _p(vo))
        return cp, ro, vo

    def csr_to_csc(self, m, rp, col, val=None):
        n, nnz = len(rp) - 1, len(col)
        cp = np.empty(m + 1, rp.dtype)
        ro = np.empty(nnz, rp.dtype)
        vo = None if val is None else np.empty_like(val)
        self.lib.orc_csr_to_csc(it_of(rp), vt_of(val), C.c_int64(n), C.c_int64(m), C.c_int64(nnz), _p(rp), _p(col),
                                _p(val), _p(cp), _p(ro), _p(vo))
        return cp, ro, vo

    def degree_reorder(self, rp, ascending=True, col=None, m=None):
        n = len(rp) - 1
        inv = np.empty(n, rp.dtype)
        self.lib.orc_degree_reorder(it_of(rp), C.c_int64(n), _p(rp), int(ascending), _p(inv))
        return inv

    def rcm_reorder(self, rp, col, with_stats=False):
        n = len(rp) - 1
        inv = np.empty(n, rp.dtype)
        stats = np.zeros(8, np.int64)
        self.lib.orc_rcm_reorder(it_of(rp), C.c_int64(n), _p(rp), _p(col), _p(inv), _p(stats))
        return (inv, stats) if with_stats else inv

    def gray_reorder(self, rp, col, m, resolution, nnz_threshold, group_size):
        n = len(rp) - 1
        inv = np.empty(n, rp.dtype)
        rc = self.lib.orc_gray_reorder(it_of(rp), C.c_int64(n), C.c_int64(m), _p(rp), _p(col),
                                       int(resolution), int(nnz_threshold), int(group_size), _p(inv))
        if rc != 0:
            raise ValueError("gray_reorder: shape undefined in the reference (m %% resolution != 0)")
        return inv

    def gray_row_keys(self, rp, col, m, resolution, nnz_threshold):
        n = len(rp) - 1
        deg = np.empty(n, rp.dtype)
        key = np.empty(n, np.uint64)
        counts = np.zeros(4, np.int64)
        rc = self.lib.orc_gray_row_keys(it_of(rp), C.c_int64(n), C.c_int64(m), _p(rp), _p(col),
                                        int(resolution), int(nnz_threshold), _p(deg), _p(key), _p(counts))
        if rc != 0:
            raise ValueError("gray_row_keys: undefined shape")
        return deg, key, counts

    def permute_csr(self, rp, col, val, row_order, col_order, m=None):
        n, nnz = len(rp) - 1, len(col)
        rpo = np.empty(n + 1, rp.dtype)
        co = np.empty(nnz, rp.dtype)
        vo = None if val is None else np.empty_like(val)
        self.lib.orc_permute_csr(it_of(rp), vt_of(val), C.c_int64(n), _p(rp), _p(col), _p(val),
                                 _p(row_order), _p(col_order), _p(rpo), _p(co), _p(vo))
        return rpo, co, vo

    def inverse_permutation(self, perm):
        inv = np.empty_like(perm)
        self.lib.orc_inverse_permutation(it_of(perm), C.c_int64(len(perm)), _p(perm), _p(inv))
        return inv

    def permute_array(self, order, vals):
        out = np.empty_like(vals)
        self.lib.orc_permute_array(it_of(order), vt_of(vals), C.c_int64(len(order)), _p(order), _p(vals), _p(out))
        return out


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libsbref.so"))


class Ref:
    """The real reference.  Only (index, value) tuples built in ref_driver.cc work."""

    def __init__(self):
        self.lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libsbref.so"))

    @staticmethod
    def _chk(rc):
        if rc != 0:
            raise TypeError("type tuple not built into oracle/_ref/libsbref.so")

    def mixed_pipeline(self, n, m, row, col, val, row_order, col_order, with_rcm):
        """The real reference instantiated for <int ids, long long offsets, float>: COO -> CSR (row_ptr int64) -> COO,
        Permute2D, DegreeReorder(ascending), RCMReorder (ref_driver.cc: ref_mixed_pipeline)."""
        nnz = len(row)
        rp = np.empty(n + 1, np.int64)
        back = np.empty(nnz, np.int32)
        prp = np.empty(n + 1, np.int64)
        pcol = np.empty(nnz, np.int32)
        pval = np.empty(nnz, np.float32)
        deg = np.empty(n, np.int32)
        rcm = np.empty(n, np.int32) if with_rcm else None
        rc = self.lib.ref_mixed_pipeline(C.c_int64(n), C.c_int64(m), C.c_int64(nnz), _p(row.copy()), _p(col.copy()),
                                         _p(val.copy()), _p(row_order.copy()), _p(col_order.copy()), _p(rp), _p(back),
                                         _p(prp), _p(pcol), _p(pval), _p(deg), _p(rcm))
        if rc != 0:
            raise RuntimeError("ref_mixed_pipeline failed: %d" % rc)
        return rp, back, (prp, pcol, pval), deg, rcm

    def coo_sort(self, row, col, val=None, n=None, m=None):
        row, col = row.copy(), col.copy()
        val = None if val is None else val.copy()
        n = int(row.max()) + 1 if n is None else n
        m = int(col.max()) + 1 if m is None else m
        self._chk(self.lib.ref_coo_sort(it_of(row), vt_of(val), C.c_int64(n), C.c_int64(m),
                                        C.c_int64(len(row)), _p(row), _p(col), _p(val)))
        return row, col, val

    def csr_sort_rows(self, rp, col, val=None, m=None):
        col = col.copy()
        val = None if val is None else val.copy()
        n = len(rp) - 1
        m = (int(col.max()) + 1 if len(col) else 1) if m is None else m
        self._chk(self.lib.ref_csr_sort_rows(it_of(rp), vt_of(val), C.c_int64(n), C.c_int64(m),
                                             _p(rp), _p(col), _p(val)))
        return col, val

    def coo_to_csr(self, n, row, col, val=None, m=None):
        nnz = len(row)
        m = (int(col.max()) + 1 if nnz else 1) if m is None else m
        rp = np.empty(n + 1, row.dtype)
        co = np.empty(nnz, row.dtype)
        vo = None if val is None else np.empty_like(val)
        self._chk(self.lib.ref_coo_to_csr(it_of(row), vt_of(val), C.c_int64(n), C.c_int64(m),
                                          C.c_int64(nnz), _p(row), _p(col), _p(val), _p(rp), _p(co), _p(vo)))
        return rp, co, vo

    def csr_to_coo(self, rp, col, val=None, m=None):
        n, nnz = len(rp) - 1, len(col)
        m = (int(col.max()) + 1 if nnz else 1) if m is None else m
        ro = np.empty(nnz, rp.dtype)
        co = np.empty(nnz, rp.dtype)
        vo = None if val is None else np.empty_like(val)
        self._chk(self.lib.ref_csr_to_coo(it_of(rp), vt_of(val), C.c_int64(n), C.c_int64(m),
                                          C.c_int64(nnz), _p(rp), _p(col), _p(val), _p(ro), _p(co), _p(vo)))
        return ro, co, vo

    def mtx_read(self, path, zero_index=True, upper=False, index_dtype=np.int32, value_dtype=None, cap=1 << 22):
        """The real MTXReader::ReadCOO on a file: (n, m, row, col, val) - already sorted by the COO constructor."""
        row, col = np.empty(cap, index_dtype), np.empty(cap, index_dtype)
        val = None if value_dtype is None else np.empty(cap, value_dtype)
        dims = (C.c_int64 * 3)()
        rc = self.lib.ref_mtx_read(0 if index_dtype == np.int32 else 1, vt_of(val), str(path).encode(), int(zero_index),
                                   int(upper), C.c_int64(cap), _p(row), _p(col), _p(val), dims)
        if rc != 0:
            raise ValueError(f"ref_mtx_read rc={rc}")
        k = dims[2]
        return dims[0], dims[1], row[:k].copy(), col[:k].copy(), (None if val is None else val[:k].copy())

    def edge_list_read(self, path, weighted=False, remove_duplicates=False, remove_self_edges=False,
                       read_undirected=True, square=False, index_dtype=np.int32, value_dtype=None, cap=1 << 22):
        row, col = np.empty(cap, index_dtype), np.empty(cap, index_dtype)
        val = None if value_dtype is None else np.empty(cap, value_dtype)
        dims = (C.c_int64 * 3)()
        rc = self.lib.ref_edge_list_read(0 if index_dtype == np.int32 else 1, vt_of(val), str(path).encode(), int(weighted),
                                         int(remove_duplicates), int(remove_self_edges), int(read_undirected), int(square),
                                         C.c_int64(cap), _p(row), _p(col), _p(val), dims)
        if rc != 0:
            raise ValueError(f"ref_edge_list_read rc={rc}")
        k = dims[2]
        return dims[0], dims[1], row[:k].copy(), col[:k].copy(), (None if val is None else val[:k].copy())

    # ---- SbFF binary containers, <int,int,float> / Array<float>
    def sbff_write_coo(self, path, n, m, row, col, vals=None):
        rc = self.lib.ref_sbff_write_coo(str(path).encode(), int(n), int(m), len(row), _p(row), _p(col), _p(vals))
        if rc != 0:
            raise ValueError(f"ref_sbff_write_coo rc={rc}")

    def sbff_write_csr(self, path, n, m, row_ptr, col, vals=None):
        rc = self.lib.ref_sbff_write_csr(str(path).encode(), int(n), int(m), _p(row_ptr), _p(col), _p(vals))
        if rc != 0:
            raise ValueError(f"ref_sbff_write_csr rc={rc}")

    def sbff_write_array(self, path, vals):
        rc = self.lib.ref_sbff_write_array(str(path).encode(), len(vals), _p(vals))
        if rc != 0:
            raise ValueError(f"ref_sbff_write_array rc={rc}")

    def sbff_read_coo(self, path, cap=1 << 20):
        row, col, vals = np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float32)
        dims = (C.c_int64 * 4)()
        rc = self.lib.ref_sbff_read_coo(str(path).encode(), C.c_int64(cap), _p(row), _p(col), _p(vals), dims)
        if rc != 0:
            raise ValueError(f"ref_sbff_read_coo rc={rc}")
        k = dims[2]
        return dims[0], dims[1], row[:k].copy(), col[:k].copy(), (vals[:k].copy() if dims[3] else None)

    def sbff_read_csr(self, path, cap=1 << 20):
        rp, col, vals = np.empty(cap + 1, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float32)
        dims = (C.c_int64 * 4)()
        rc = self.lib.ref_sbff_read_csr(str(path).encode(), C.c_int64(cap), C.c_int64(cap), _p(rp), _p(col), _p(vals), dims)
        if rc != 0:
            raise ValueError(f"ref_sbff_read_csr rc={rc}")
        k = dims[2]
        return dims[0], dims[1], rp[:dims[0] + 1].copy(), col[:k].copy(), (vals[:k].copy() if dims[3] else None)

    def sbff_read_array(self, path, cap=1 << 20):
        vals = np.empty(cap, np.float32)
        n = C.c_int64(0)
        rc = self.lib.ref_sbff_read_array(str(path).encode(), C.c_int64(cap), _p(vals), C.byref(n))
        if rc != 0:
            raise ValueError(f"ref_sbff_read_array rc={rc}")
        return vals[:n.value].copy()

    def features(self, rp, col):
        """(bandwidth, profile as IDType, degrees, float distribution, double distribution) of a square CSR."""
        n = len(rp) - 1
        bw, pf = C.c_int64(0), C.c_int64(0)
        deg = np.empty(n, rp.dtype)
        df, dd = np.empty(n, np.float32), np.empty(n, np.float64)
        self._chk(self.lib.ref_features(it_of(rp), C.c_int64(n), _p(rp), _p(col), C.byref(bw), C.byref(pf), _p(deg),
                                        _p(df), _p(dd)))
        return bw.value, pf.value, deg, df, dd

    def coo_to_csc(self, n, m, row, col, val=None):
        assert n == m, "the reference's COO->CSC is only memory-safe for square matrices"
        nnz = len(row)
        cp = np.empty(n + 1, row.dtype)
        ro = np.empty(nnz, row.dtype)
        vo = None if val is None else np.empty_like(val)
        self._chk(self.lib.ref_coo_to_csc(it_of(row), vt_of(val), C.c_int64(n), C.c_int64(nnz), _p(row), _p(col),
                                          _p(val), _p(cp), _p(ro), _p(vo)))
        return cp, ro, vo

    def csr_to_csc(self, m, rp, col, val=None):
        n, nnz = len(rp) - 1, len(col)
        assert n == m, "the reference's CSR->CSC is only memory-safe for square matrices"
        cp = np.empty(n + 1, rp.dtype)
        ro = np.empty(nnz, rp.dtype)
        vo = None if val is None else np.empty_like(val)
        self._chk(self.lib.ref_csr_to_csc(it_of(rp), vt_of(val), C.c_int64(n), C.c_int64(nnz), _p(rp), _p(col),
                                          _p(val), _p(cp), _p(ro), _p(vo)))
        return cp, ro, vo

    def degree_reorder(self, rp, ascending=True, col=None, m=None):
        n = len(rp) - 1
        if col is None:
            col = np.zeros(int(rp[-1]), rp.dtype)
        m = n if m is None else m
        inv = np.empty(n, rp.dtype)
        self._chk(self.lib.ref_degree_reorder(it_of(rp), V_I32 if it_of(rp) == 0 else V_NONE, C.c_int64(n),
                                              C.c_int64(m), _p(rp), _p(col), int(ascending), _p(inv)))
        return inv

    def rcm_reorder(self, rp, col):
        n = len(rp) - 1
        inv = np.empty(n, rp.dtype)
        vt = V_U32 if rp.dtype == np.uint32 else (V_I32 if it_of(rp) == 0 else V_NONE)
        self._chk(self.lib.ref_rcm_reorder(it_of(rp), vt, C.c_int64(n), _p(rp), _p(col), _p(inv)))
        return inv

    def gray_reorder(self, rp, col, m, resolution, nnz_threshold, group_size):
        n = len(rp) - 1
        inv = np.empty(n, rp.dtype)
        self._chk(self.lib.ref_gray_reorder(it_of(rp), V_I32 if it_of(rp) == 0 else V_NONE, C.c_int64(n),
                                            C.c_int64(m), _p(rp), _p(col), int(resolution),
                                            int(nnz_threshold), int(group_size), _p(inv)))
        return inv

    def permute_csr(self, rp, col, val, row_order, col_order, m=None):
        n, nnz = len(rp) - 1, len(col)
        m = (int(col.max()) + 1 if nnz else 1) if m is None else m
        if row_order is None:
            # the reference deletes an uninitialised pointer when row_order is null
            # (permute/permute_order_two.cc:75); drive it with an explicit identity.
            row_order = np.arange(n, dtype=rp.dtype)
        rpo = np.empty(n + 1, rp.dtype)
        co = np.empty(nnz, rp.dtype)
        vo = None if val is None else np.empty_like(val)
        self._chk(self.lib.ref_permute_csr(it_of(rp), vt_of(val), C.c_int64(n), C.c_int64(m), _p(rp),
                                           _p(col), _p(val), _p(row_order), _p(col_order), _p(rpo),
                                           _p(co), _p(vo)))
        return rpo, co, vo
