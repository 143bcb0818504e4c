#!/bin/bash
# The 14 - 22 ms between submission and start of the Gray key stage inside reorder_cli (tools/gray_kt2.sh): which host-side
# circumstance causes it?  Same call, four ways: as shipped (page-locked staging), pageable staging, pageable + the host
# busy for 30 / 100 ms between the calls.
N=$((1 << 22))
python3 -c "
import sys; sys.path.insert(0, '.')
from sparsebase_amd import synth
rp, col = (t.cpu().numpy() for t in synth.banded_symmetric_torch($N, 64, per_row=12, seed=2))
rp.tofile('/tmp/g_rp.bin'); col.tofile('/tmp/g_col.bin')
rp, col = (t.cpu().numpy() for t in synth.rmat_symmetric_torch(22, 13, seed=1))
rp.tofile('/tmp/r_rp.bin'); col.tofile('/tmp/r_col.bin')"
CLI=sparsebase_amd/host/bin/reorder_cli
for m in g r; do
for rep in 1 2 3; do
  echo "== matrix $m rep $rep: pinned staging";            $CLI gray /tmp/${m}_rp.bin /tmp/${m}_col.bin /tmp/g_out.bin $N $N 32 10 4 --device --time 2>/dev/null | tr '\n' ' '; echo
  echo "== pageable staging";          SBX_HOST_PINNED_STAGING=0 $CLI gray /tmp/${m}_rp.bin /tmp/${m}_col.bin /tmp/g_out.bin $N $N 32 10 4 --device --time 2>/dev/null | tr '\n' ' '; echo
done
  echo "== pageable, 30 ms busy host in front";  SBX_HOST_PINNED_STAGING=0 $CLI gray /tmp/${m}_rp.bin /tmp/${m}_col.bin /tmp/g_out.bin $N $N 32 10 4 --device --time --idle-ms 30 2>/dev/null | tr '\n' ' '; echo
  echo "== pageable, 100 ms busy host in front"; SBX_HOST_PINNED_STAGING=0 $CLI gray /tmp/${m}_rp.bin /tmp/${m}_col.bin /tmp/g_out.bin $N $N 32 10 4 --device --time --idle-ms 100 2>/dev/null | tr '\n' ' '; echo
  echo "== pinned, 30 ms busy host in front";  $CLI gray /tmp/${m}_rp.bin /tmp/${m}_col.bin /tmp/g_out.bin $N $N 32 10 4 --device --time --idle-ms 30 2>/dev/null | tr '\n' ' '; echo
done
