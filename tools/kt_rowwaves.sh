#!/bin/bash
# row-class kernel durations against the resident waves per CU the grids are sized for (SBX_PERMUTE_ROW_WAVES)
export SBX_PROBE_LIB=${SBX_PROBE_LIB:-tuning}
for wv in 4 8 12 16 24 32; do
  SBX_PERMUTE_ROW_WAVES=$wv KT_N=40 tools/kt_permute.sh rw$wv "$@" > /dev/null
  echo "== waves/CU $wv"; grep -E "k_rows_quad" gpurun_out/kt_rw$wv.txt | head -5
done
