#!/bin/bash
# N concurrent tools/rcm_shared_loop.py on one GPU.  usage: tools/rcm_shared_loop.sh [N] [rounds]
N=${1:-8}; R=${2:-50}
for i in $(seq 1 $N); do python tools/rcm_shared_loop.py $R > /tmp/rcm_loop_$i.log 2>&1 & done
wait
cat /tmp/rcm_loop_*.log | grep -E "MISMATCH|EXC|loop:" | sort | uniq -c | sort -rn | head -20
