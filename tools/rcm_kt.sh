#!/bin/bash
# kernel trace of one warm RCM on the bench matrix -> gpurun_out/rcm_timeline_<tag>.txt (tools/rcm_timeline.py --all)
TAG=${1:-x}
export TMPDIR=/tmp
rm -rf /tmp/rcm_kt_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/rcm_kt_$TAG -o kt -- python3 tools/rcm_trace.py > /dev/null 2>&1
python3 tools/rcm_timeline.py /tmp/rcm_kt_$TAG --all > gpurun_out/rcm_timeline_$TAG.txt 2>&1
tail -1 gpurun_out/rcm_timeline_$TAG.txt
