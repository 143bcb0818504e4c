#!/bin/bash
# row-class kernels: everything / no sort (rows stream out unsorted) / no relabel gathers (and therefore no sort) — timing ablation
for f in 0 2 4 6; do
  SBX_PERMUTE_ROW_WAVES=${RW:-8} SBX_PERMUTE_FORCE_RADIX=$f KT_N=40 tools/kt_permute.sh ab$f "$@" > /dev/null
  echo "== force_radix $f (2: no sort, 4: no gathers, 6: neither)"; grep -E "k_rows_quad|k_permute_block_rows<int, 4, (256|512|1024)," gpurun_out/kt_ab$f.txt | head -6
done
