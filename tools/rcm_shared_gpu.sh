#!/bin/bash
# N concurrent processes on one GPU, each checking RCM against the oracle (tools/rcm_modes.py): grid barriers give up,
# sweeps fall back — the results must not change.  usage: tools/rcm_shared_gpu.sh [N]
N=${1:-8}
for i in $(seq 1 $N); do python tools/rcm_modes.py > /tmp/rcm_shared_$i.log 2>&1 & done
wait
cat /tmp/rcm_shared_*.log | grep -E "MISMATCH|EXC|modes" | sort | uniq -c
