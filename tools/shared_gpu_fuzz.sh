#!/bin/bash
# N concurrent differential fuzzers (tools/fuzz_ops.py) on one GPU: every operation against the oracle while the
# kernels of other processes interleave.  usage: tools/shared_gpu_fuzz.sh [N] [rounds]
N=${1:-8}; R=${2:-150}
for i in $(seq 1 $N); do python tools/fuzz_ops.py $R $((100 + i)) > /tmp/fuzz_shared_$i.log 2>&1 & done
wait
tail -q -n 1 /tmp/fuzz_shared_*.log | sort | uniq -c
grep -l "MISMATCH\|Traceback" /tmp/fuzz_shared_*.log | head
