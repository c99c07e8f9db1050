#!/bin/bash
# A/B: grids of a sub-tile contiguous in LDS (no 16 bytes between them, no per-piece division when they are staged;
# tools/libmgrid_nopad.so: -DMG_ONLY_VIEW7 -DMG_NO_GRID_PAD) against the product build.  Two rounds.
for round in 1 2; do
for l in "" tools/libmgrid_nopad.so; do
  echo "== round $round ${l:-product}"
  if [ -z "$l" ]; then unset MGRID_LIB; else export MGRID_LIB=$l; fi
  bash tools/exp_quick.sh "3 1" "4 1" "5 1" "2 1" "5 8" "3 8" 2>&1 | grep -v amdgpu.ids
done
done
