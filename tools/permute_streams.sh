#!/bin/bash
# Permute2D end to end under different assignments of the row classes to side streams (SBX_PERMUTE_CLASS_STREAMS)
for rep in 1 2 3; do for m in 000000 023456 002222 000022; do
  echo -n "streams=$m: "; SBX_PERMUTE_CLASS_STREAMS=$m python tools/permute_time.py | tail -2 | tr '\n' ' '; echo
done; done
