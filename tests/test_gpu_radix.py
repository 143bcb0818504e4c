"""The library's radix sort itself (internal entry points, reached through their mangled names): every record shape,
counts around the tile sizes of the persistent pass kernel, 1 ... 8 digit passes, against torch's stable sort."""
import ctypes as C
import re
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class Pass(C.Structure):
    _fields_ = [("shift", C.c_int), ("bits", C.c_int)]


@pytest.fixture(scope="module")
def rs():
    from sparsebase_amd import capi, ops
    hd = ops.handle_for(torch.device("cuda", 0))
    names = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True).stdout

    def sym(stem):
        m = re.search(r"\b(_Z\d+%s\w*)" % stem, names)
        assert m, f"{stem} is not exported by {capi.LIB_PATH}"
        f = getattr(hd.lib, m.group(1))
        f.restype = C.c_int
        return f

    return hd, sym("sbx_radix_sortP"), sym("sbx_radix_plan"), sym("sbx_arena_begin")


def ptr(t):
    return C.c_void_p(t.data_ptr() if t is not None else 0)


@pytest.mark.parametrize("kbytes,pbytes", [(4, 0), (4, 4), (4, 8), (8, 0), (8, 4), (8, 8)])
def test_radix_sort_record_shapes_and_tile_edges(rs, kbytes, pbytes):
    hd, sort, plan, arena_begin = rs
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1000 * kbytes + pbytes)
    tile = 512 * (4 if (kbytes, pbytes) == (8, 8) else 8)
    counts = [2, 63, 64, 65, tile - 1, tile, tile + 1, 2 * tile - 1, 2 * tile + 1, 3 * tile, 100_003, 1_500_000]
    # (low bits, high bits): 1 pass ... 8 passes; narrow ranges give long runs of equal keys (stability)
    ranges = [(3, 0), (8, 0), (9, 0), (20, 0), (31, 0)] if kbytes == 4 else [(1, 1), (7, 9), (20, 20), (31, 31), (32, 32)]
    for count in counts:
        for lo_bits, hi_bits in ranges:
            keys = torch.randint(0, 1 << lo_bits, (count,), device=dev, dtype=torch.int64, generator=g)
            if kbytes == 8:
                keys = keys | (torch.randint(0, 1 << hi_bits, (count,), device=dev, dtype=torch.int64, generator=g) << 32)
            else:
                keys = keys.to(torch.int32)
            pay = None
            if pbytes:
                pay = torch.arange(count, device=dev, dtype=torch.int64 if pbytes == 8 else torch.int32) * 3 + 1
            passes = (Pass * 16)()
            n_passes = plan(0, lo_bits, 32, 32 + hi_bits, passes)
            assert 1 <= n_passes <= 8
            ka, kb = keys.clone(), torch.zeros_like(keys)
            pa = pay.clone() if pbytes else None
            pb = torch.zeros_like(pay) if pbytes else None
            in_b = C.c_int(0)
            hd.bind_stream()
            hd.check(arena_begin(hd.h))
            hd.check(sort(hd.h, kbytes, pbytes, ptr(ka), ptr(kb), ptr(pa), ptr(pb), C.c_int64(count), passes, n_passes,
                          C.byref(in_b)))
            torch.cuda.synchronize()
            # unsigned order == signed order here: the keys are non-negative except (32, 32), compared as unsigned
            ref_keys = keys if not (kbytes == 8 and hi_bits == 32) else keys ^ (1 << 63)
            order = torch.sort(ref_keys, stable=True)[1]
            got_k = kb if in_b.value else ka
            assert torch.equal(got_k, keys[order]), (count, lo_bits, hi_bits)
            if pbytes:
                got_p = pb if in_b.value else pa
                assert torch.equal(got_p, pay[order]), (count, lo_bits, hi_bits)
