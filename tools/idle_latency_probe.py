#!/usr/bin/env python3
"""How long does a small call take after the GPU has been idle for x ms?  (The exact GrayReorder's device stage reads
10 - 20 ms inside a real call while its kernels need 0.1 - 0.26 ms: tools/gray_kt2.sh shows the kernels START ~14 ms after
they were submitted.)  Sweeps the idle time in front of one sbx_gray_row_keys call on the banded C5 matrix, with and
without a 48 MB pageable D2H copy in front of the idle period (what the Gray call does)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sparsebase_amd import ops, synth
n = 1 << 22
rp, col = synth.banded_symmetric_torch(n, 64, per_row=12, seed=2)
for _ in range(3):
    deg, key, _ = ops.gray_row_keys(n, rp, col, 32, 10)
torch.cuda.synchronize()
for copy in (False, True):
    for idle_ms in (0, 0.5, 1, 2, 4, 6, 8, 10, 12, 15, 20, 30, 50, 80, 150):
        ts = []
        for rep in range(4):
            deg, key, _ = ops.gray_row_keys(n, rp, col, 32, 10)
            if copy:
                d, k = deg.cpu(), key.cpu()
            torch.cuda.synchronize()
            t_end = time.perf_counter() + idle_ms * 1e-3
            while time.perf_counter() < t_end:   # (busy wait: the host stage computes, it does not sleep)
                pass
            t = time.perf_counter()
            ops.gray_row_keys(n, rp, col, 32, 10)
            ts.append((time.perf_counter() - t) * 1e3)
        print(f"copy={int(copy)} idle {idle_ms:6.1f} ms -> call " + " ".join(f"{x:7.3f}" for x in ts) + " ms", flush=True)
