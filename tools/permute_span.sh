#!/bin/bash
# one Permute2D with its side streams on, as a kernel-trace timeline -> gpurun_out/permute_span_<tag>.txt
TAG=${1:-x}; shift
export TMPDIR=/tmp
rm -rf /tmp/ps_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/ps_$TAG -o kt -- python3 tools/permute_only.py "$@" > /dev/null 2>&1
python3 tools/permute_span.py /tmp/ps_$TAG > gpurun_out/permute_span_$TAG.txt 2>&1
tail -1 gpurun_out/permute_span_$TAG.txt
