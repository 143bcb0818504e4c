#!/usr/bin/env python3
"""One RCM on the bench matrix (after a warm-up call): the target of `rocprofv3 --kernel-trace` when the per-launch
timeline of a single call is wanted (tools/trace_summary.py prints it)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparsebase_amd import ops, synth
rp, col = synth.rmat_symmetric_torch(22, 13, seed=1)
ops.rcm_reorder(rp, col)
torch.cuda.synchronize()
ops.rcm_reorder(rp, col)
torch.cuda.synchronize()
