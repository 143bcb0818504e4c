"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every
symbol include/sbx.h declares, the ctypes table matches the header, and creating a handle
without a device fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from sparsebase_amd import build
    return build.build()


def header_functions():
    text = open(os.path.join(ROOT, "include", "sbx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sbx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    fns = header_functions()
    for must in ("sbx_coo_to_csr", "sbx_csr_to_coo", "sbx_coo_sort", "sbx_csr_sort_rows", "sbx_degree_reorder",
                 "sbx_rcm_reorder", "sbx_gray_row_keys", "sbx_permute_csr", "sbx_permute_csr_rows",
                 "sbx_inverse_permutation", "sbx_permute_array"):
        assert must in fns


def test_library_exports_every_declared_symbol(lib_path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path], text=True)
    exported = set(re.findall(r"\bT (sbx_[a-z0-9_]+)", out))
    missing = [f for f in header_functions() if f not in exported]
    assert not missing, f"declared in include/sbx.h but not exported: {missing}"


def test_ctypes_table_matches_header(lib_path):
    from sparsebase_amd import capi
    assert sorted(capi.PROTOTYPES) == header_functions()
    lib = capi.load()
    assert lib.sbx_version() == 102
    assert lib.sbx_status_string(2) == b"no usable HIP device"
    names = [lib.sbx_profile_kernel_name(i).decode() for i in range(lib.sbx_profile_kernel_count())]
    assert "permute_tile" in names and "bfs_expand" in names


def test_no_device_fails_loudly(lib_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from sparsebase_amd import capi
    lib = capi.load()
    cnt = C.c_int(-1)
    assert lib.sbx_device_count(C.byref(cnt)) == 2 and cnt.value == 0  # SBX_ERR_NO_DEVICE
    h = C.c_void_p()
    assert lib.sbx_create(0, C.byref(h)) == 2 and not h.value
    # the tensor layer refuses CPU tensors instead of computing on the host
    from sparsebase_amd import ops
    with pytest.raises(ValueError):
        ops.degree_reorder(torch.zeros(4, dtype=torch.int32))


def test_product_does_not_touch_the_oracle():
    # nothing under sparsebase_amd/ may import, link or execute oracle/ (it is test infrastructure)
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sparsebase_amd")):
        if os.sep + "lib" in dirpath or os.sep + "bin" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cc", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"sbx_oracle|libsbref|orc_[a-z_]+\(|oracle/", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
