#!/bin/bash
# Collects the rocprofv3 evidence for profiles/ on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the bench command, then separate FETCH_SIZE / WRITE_SIZE counter passes.
# usage: tools/collect_profiles.sh <tag>     -> gpurun_out/<tag>_{kernel_stats.csv,pmc_traffic.json,bench_line.json}
set -u
TAG=${1:-r1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
# per-kernel durations are taken with the three permute paths serialised (as bench.py's instrumented pass does):
# overlapped on side streams, a kernel's duration includes its neighbours' share of the machine
export SBX_PERMUTE_OVERLAP=0
export SBX_RCM_OVERLAP=0
ARGS="bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-sharded"
cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- python3 $ARGS > "$OUT/${TAG}_kt_bench.log" 2>&1
cp $(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1) "$OUT/${TAG}_kernel_stats.csv"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o f -- python3 $ARGS > "$OUT/${TAG}_pmc_f.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o w -- python3 $ARGS > "$OUT/${TAG}_pmc_w.log" 2>&1
python3 tools/summarize_pmc.py /tmp/prof_f /tmp/prof_w "$OUT/${TAG}_pmc_traffic.json" > /dev/null
SBX_PERMUTE_OVERLAP=1 SBX_RCM_OVERLAP=1 python3 bench.py --steps 10 --warmup 2 > "$OUT/${TAG}_bench_line.json" 2> "$OUT/${TAG}_bench_line.err"
tail -c 600 "$OUT/${TAG}_bench_line.json"
