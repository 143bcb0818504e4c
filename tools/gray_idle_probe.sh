#!/bin/bash
# RMAT-22 through reorder_cli gray --device --time, N runs with and without a busy-wait in front of the timed call
python3 - <<PY
import sys; sys.path.insert(0, "/root/repo")
from sparsebase_amd import synth
rp, col = (t.cpu().numpy() for t in synth.rmat_symmetric_torch(22, 13, seed=1))
rp.tofile("/tmp/g_rp.bin"); col.tofile("/tmp/g_col.bin")
PY
CLI=sparsebase_amd/host/bin/reorder_cli
for idle in 0 300 1000; do
  echo "idle $idle ms"
  for i in 1 2 3 4 5 6 7 8; do
    timeout 120 $CLI gray /tmp/g_rp.bin /tmp/g_col.bin /tmp/g_out.bin 4194304 4194304 32 10 4 --device --time --idle-ms $idle 2>&1 | tr '\n' ' ' | cut -c1-200; echo
  done
done
