#!/usr/bin/env python3
"""Builds a variant of the library for A/B probes: tools/build_variant.py <name> <file.hip> -DX=1 ... compiles that one
source with the extra flags and links sparsebase_amd/lib/libsbx_<name>.so from it and the product's other objects.
Probes pick it up through SBX_PROBE_LIB=<name>."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebase_amd import build as B
name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()
obj = os.path.join(B.OBJDIR, f"{src[:-4]}_{name}.o")
subprocess.check_call([B.HIPCC] + B.FLAGS + extra + ["-c", os.path.join(B.CSRC, src), "-o", obj])
objs = [os.path.join(B.OBJDIR, os.path.basename(s)[:-4] + ".o") for s in B._sources() if not s.endswith("/" + src)]
lib = os.path.join(B.LIBDIR, f"libsbx_{name}.so")
subprocess.check_call([B.HIPCC, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", lib] + objs + [obj, "-ldl"])
print(lib)
