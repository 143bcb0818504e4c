#!/usr/bin/env python3
"""Phase times inside k_onesweep_pass.  `--build` (here, no GPU needed) compiles sbx_prims.hip with -DSBX_RADIX_STAMPS
and links sparsebase_amd/lib/libsbx_stamps.so next to the product library; without it (on the GPU box) the probe runs
one radix sort through that library and prints the average cycles (100 MHz clock -> us) a tile spends up to each stamp."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebase_amd import build as B
LIB = os.path.join(B.LIBDIR, "libsbx_stamps.so")
if "--build" in sys.argv:
    B.build()
    obj = os.path.join(B.OBJDIR, "sbx_prims_stamps.o")
    subprocess.check_call([B.HIPCC] + B.FLAGS + ["-DSBX_RADIX_STAMPS", "-c", os.path.join(B.CSRC, "sbx_prims.hip"), "-o", obj])
    objs = [os.path.join(B.OBJDIR, os.path.basename(s)[:-4] + ".o") for s in B._sources() if not s.endswith("sbx_prims.hip")]
    subprocess.check_call([B.HIPCC, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", LIB] + objs + [obj, "-ldl"])
    print(LIB)
    sys.exit(0)
import numpy as np, torch
from sparsebase_amd import capi
capi.LIB_PATH = LIB
from sparsebase_amd import ops
dev = torch.device("cuda", 0)
hd = ops.handle_for(dev)
lib = hd.lib
fn = getattr(lib, "_Z14sbx_radix_sortP12sbx_handle_siiPvS1_S1_S1_lPK14sbx_radix_passiPi"); fn.restype = C.c_int
plan = getattr(lib, "_Z14sbx_radix_planiiiiP14sbx_radix_pass"); plan.restype = C.c_int
arena_begin = getattr(lib, "_Z15sbx_arena_beginP12sbx_handle_s"); arena_begin.restype = C.c_int
class Pass(C.Structure):
    _fields_ = [("shift", C.c_int), ("bits", C.c_int)]
count = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
keys = torch.randint(0, 1 << 20, (count,), device=dev, dtype=torch.int64).to(torch.int32)
pay = torch.arange(count, device=dev, dtype=torch.int64)
kb, pb = torch.empty_like(keys), torch.empty_like(pay)
passes = (Pass * 16)()
np_ = plan(0, 20, 0, 0, passes)
in_b = C.c_int(0)
st = (C.c_ulonglong * 16)()
for r in range(3):
    ka, pa = keys.clone(), pay.clone()
    hd.bind_stream(); hd.check(arena_begin(hd.h))
    torch.cuda.synchronize()
    lib.sbx_debug_radix_stamps(st, 1)
    hd.check(fn(hd.h, 4, 8, C.c_void_p(ka.data_ptr()), C.c_void_p(kb.data_ptr()), C.c_void_p(pa.data_ptr()),
                C.c_void_p(pb.data_ptr()), C.c_int64(count), passes, np_, C.byref(in_b)))
    torch.cuda.synchronize()
    lib.sbx_debug_radix_stamps(st, 0)
tiles = st[15]
names = ["ticket+clear", "loads landed", "ranked", "block scans", "look-back", "LDS scatter", "stores issued", "stores done"]
prev = 0.0
for i, nm in enumerate(names):
    us = st[i] / tiles / 100.0
    print(f"{nm:>14}: {us:7.2f} us  (+{us - prev:6.2f})")
    prev = us
print("tiles (all passes):", tiles)
print(f"look-back of digit 0, per tile: {st[8] / tiles:.1f} rounds, {st[9] / tiles:.1f} of them stalled, {st[10] / tiles:.1f} predecessors read")
