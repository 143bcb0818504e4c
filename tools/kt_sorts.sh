#!/bin/bash
# Kernel timelines of the sort-based conversions: the COO constructor sort of C2B and CSR -> CSC of C2.
# usage (gpurun, repo root): tools/kt_sorts.sh  -> gpurun_out/sort_timeline_{c2b,csc}.txt
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
rm -rf /tmp/kts_a /tmp/kts_b
COO_PROBE_ONLY=c2b rocprofv3 --kernel-trace --output-format csv -d /tmp/kts_a -o t -- python3 tools/coo_sort_probe.py > "$OUT/kt_sorts_a.log" 2>&1
python3 tools/op_timeline.py /tmp/kts_a k_coo_is_sorted > "$OUT/sort_timeline_c2b.txt"
rocprofv3 --kernel-trace --output-format csv -d /tmp/kts_b -o t -- python3 tools/csc_once.py > "$OUT/kt_sorts_b.log" 2>&1
python3 tools/op_timeline.py /tmp/kts_b "${CSC_FIRST:-k_ex_tile_spans}" > "$OUT/sort_timeline_csc.txt"
cat "$OUT/sort_timeline_c2b.txt" "$OUT/sort_timeline_csc.txt"
