#!/bin/bash
# per-kernel durations of one Permute2D on the bench matrix (rocprofv3 kernel trace; classes serialised)
# usage: tools/kt_permute.sh <tag> [--rcm]   -> gpurun_out/kt_<tag>.txt
TAG=$1; shift
export TMPDIR=/tmp SBX_PERMUTE_OVERLAP=0
rm -rf /tmp/kt_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$TAG -o kt -- python3 tools/permute_only.py "$@" > /dev/null 2>&1
python3 tools/trace_summary.py /tmp/kt_$TAG k_permute,k_rows_quad,k_short,k_long,k_classify,k_rowwise,k_tile_first,k_fix,k_onesweep ${KT_N:-26} > gpurun_out/kt_$TAG.txt
cat gpurun_out/kt_$TAG.txt
