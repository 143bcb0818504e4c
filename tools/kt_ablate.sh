#!/bin/bash
# sort stage of the permute, per kernel: everything / no sort (rows stream out unsorted) / sort only (columns arrive
# relabelled: the same keys, no gather issued) / neither — timing ablation.  usage: tools/kt_ablate.sh [--rcm]
# (the ablation bits and SBX_PERMUTE_ROW_WAVES are live in the tuning build only: python -m sparsebase_amd.build --tuning)
export SBX_PROBE_LIB=${SBX_PROBE_LIB:-tuning}
PAT="k_permute_tile<|k_rows_quad|k_permute_block_rows<int, 4, (256|512|1024),"
for mode in "0:full" "2:nosort" "4p:sortonly" "6:neither"; do
  f=${mode%%:*}; tag=${mode##*:}; extra=""
  if [ "$f" = "4p" ]; then f=4; extra="--prerelabel"; fi
  SBX_PERMUTE_ROW_WAVES=${RW:-16} SBX_PERMUTE_FORCE_RADIX=$f KT_N=40 tools/kt_permute.sh ab_$tag "$@" $extra > /dev/null
  echo "== $tag (SBX_PERMUTE_FORCE_RADIX=$f $extra)"; grep -E "$PAT" gpurun_out/kt_ab_$tag.txt | head -8
done
