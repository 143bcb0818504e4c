#!/bin/bash
# rocprofv3 evidence for the per-operation table (tools/ops_table.py): kernel durations and, from separate
# FETCH_SIZE / WRITE_SIZE counter passes, the HBM bytes each kernel really moved -> achieved HBM GB/s.
# usage (through gpurun, repo root): tools/collect_ops_profiles.sh <tag>  -> gpurun_out/<tag>_ops_{kernel_stats.csv,hbm.json}
set -u
TAG=${1:-r1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
export SBX_PERMUTE_OVERLAP=0  # clean per-kernel durations (see collect_profiles.sh)
export SBX_RCM_OVERLAP=0
rm -rf /tmp/ops_kt /tmp/ops_f /tmp/ops_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ops_kt -o kt -- python3 tools/ops_table.py --gpu-only > "$OUT/${TAG}_ops_kt.log" 2>&1
cp $(find /tmp/ops_kt -name "*kernel_stats.csv" | head -1) "$OUT/${TAG}_ops_kernel_stats.csv"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/ops_f -o f -- python3 tools/ops_table.py --gpu-only > "$OUT/${TAG}_ops_pmc_f.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/ops_w -o w -- python3 tools/ops_table.py --gpu-only > "$OUT/${TAG}_ops_pmc_w.log" 2>&1
python3 tools/summarize_ops_hbm.py "$OUT/${TAG}_ops_kernel_stats.csv" /tmp/ops_f /tmp/ops_w "$OUT/${TAG}_ops_hbm.json"
