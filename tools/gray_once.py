#!/usr/bin/env python3
"""Two Gray key stages on the power-law bench matrix: the target of rocprofv3 --kernel-trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparsebase_amd import capi, ops, synth
if os.environ.get("SBX_PROBE_LIB"):  # a variant built by tools/build_variant.py
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{os.environ['SBX_PROBE_LIB']}.so")
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
n = rp.numel() - 1
for _ in range(2):
    ops.gray_row_keys(n, rp, col, 32, 10)
torch.cuda.synchronize()
