#!/bin/bash
# Permute2D end to end (tools/permute_time.py) under a few settings of the tuning knobs
for cfg in "" "SBX_PERMUTE_OVERLAP=0" "SBX_PERMUTE_ROW_WAVES=8" "SBX_PERMUTE_ROW_WAVES=12" "SBX_PERMUTE_BIG=1" "SBX_PERMUTE_QUAD_ROWS=0" "SBX_PERMUTE_TILE_GRID=6" "SBX_PERMUTE_TILE_GRID=24"; do
  echo "== $cfg"; env $cfg python tools/permute_time.py 2>&1 | tail -2
done
