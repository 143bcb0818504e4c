#!/bin/bash
# the multi-process tests of tests/test_sharded_gpu.py several times in a row on the final build -> gpurun_out/r6_soak.log
N=${1:-5}
: > gpurun_out/r6_soak.log
for i in $(seq 1 $N); do
  python -m pytest tests/test_sharded_gpu.py -x -q -m gpu -k "eight or fenced" 2>&1 | tail -1 >> gpurun_out/r6_soak.log
done
cat gpurun_out/r6_soak.log
