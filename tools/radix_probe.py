#!/usr/bin/env python3
"""Time the library's radix sort alone (internal entry point, reached by its mangled name): per-pass time and GB/s for the
two record shapes the converters use — (u32 key, u64 payload) over 20 key bits (CSR->CSC) and (u64 key, u32 payload)
over 40 key bits (COO constructor sort) — at 10 M and 100 M records; results checked against torch.sort."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebase_amd import capi
if os.environ.get("SBX_PROBE_LIB"):  # a variant built by tools/build_variant.py
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
from sparsebase_amd import ops

dev = torch.device("cuda", 0)
hd = ops.handle_for(dev)
lib = hd.lib
fn = getattr(lib, "_Z14sbx_radix_sortP12sbx_handle_siiPvS1_S1_S1_lPK14sbx_radix_passiPi")
fn.restype = C.c_int
plan = getattr(lib, "_Z14sbx_radix_planiiiiP14sbx_radix_pass")
plan.restype = C.c_int
arena_begin = getattr(lib, "_Z15sbx_arena_beginP12sbx_handle_s")
arena_begin.restype = C.c_int


class Pass(C.Structure):
    _fields_ = [("shift", C.c_int), ("bits", C.c_int)]


def ptr(t):
    return C.c_void_p(t.data_ptr() if t is not None else 0)


def run(count, kbytes, pbytes, bits_lo, bits_hi, reps=5):
    g = torch.Generator(device=dev); g.manual_seed(7)
    if kbytes == 4:
        keys = torch.randint(0, 1 << bits_lo, (count,), device=dev, dtype=torch.int64, generator=g).to(torch.int32)
    else:
        keys = torch.randint(0, 1 << bits_lo, (count,), device=dev, dtype=torch.int64, generator=g) | (
            torch.randint(0, 1 << bits_hi, (count,), device=dev, dtype=torch.int64, generator=g) << 32)
    if SORTED:  # every tile then writes one contiguous run per pass: isolates the cost of the scattered stores
        keys = torch.sort(keys)[0]
    pay = torch.arange(count, device=dev, dtype=torch.int64 if pbytes == 8 else torch.int32) if pbytes else None
    passes = (Pass * 16)()
    np_ = plan(0, bits_lo, 32, 32 + bits_hi, passes)
    ka, kb = keys.clone(), torch.empty_like(keys)
    pa = pay.clone() if pbytes else None
    pb = torch.empty_like(pay) if pbytes else None
    in_b = C.c_int(0)
    ts = []
    for r in range(reps + 2):
        ka.copy_(keys)
        if pbytes: pa.copy_(pay)
        hd.bind_stream()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        hd.check(arena_begin(hd.h))  # as every public entry point does: scratch is handed out from the start again
        a.record()
        hd.check(fn(hd.h, kbytes, pbytes, ptr(ka), ptr(kb), ptr(pa), ptr(pb), C.c_int64(count), passes, np_, C.byref(in_b)))
        b.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    rk = kb if in_b.value else ka
    sk, si = torch.sort(keys, stable=True)
    ok = bool(torch.equal(rk, sk))
    if pbytes:
        rp = pb if in_b.value else pa
        ok = ok and bool(torch.equal(rp, pay[si]))
    per_pass = 2 * count * (kbytes + pbytes)
    print(f"count {count:>10} key {kbytes} payload {pbytes} bits {bits_lo}+{bits_hi}: {np_} passes "
          f"{[(passes[i].shift, passes[i].bits) for i in range(np_)]} {ms:.3f} ms  "
          f"{ms / np_ * 1e3:.1f} us/pass (hist included)  {per_pass * np_ / ms / 1e6:.0f} GB/s  ok={ok}", flush=True)


SORTED = "--sorted" in sys.argv
if SORTED: sys.argv.remove("--sorted")
sizes = [int(s) for s in sys.argv[1:]] or [10_000_000, 100_000_000]
for cnt in sizes:
    run(cnt, 4, 8, 20, 0)
    run(cnt, 8, 4, 20, 20)
    run(cnt, 4, 4, 22, 0)
    run(cnt, 8, 0, 22, 22)
