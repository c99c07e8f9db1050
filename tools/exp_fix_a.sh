#!/bin/bash
# Upper bound of what kernel forms with a compile-time agent count could win: experiment builds in which KParams::A()
# and NE() are constants (-DMG_FIX_A=2 / 4 / 8: tools/libmgrid_a2.so ...), each against the product build on the config
# whose shape it matches.  Two rounds.
for round in 1 2; do
  for w in "3 2" "4 4" "5 8"; do
    set -- $w
    for l in "" tools/libmgrid_a$2.so; do
      if [ -z "$l" ]; then unset MGRID_LIB; else export MGRID_LIB=$l; fi
      echo "== round $round config $1 ${l:-product}"
      bash tools/exp_quick.sh "$1 1" "$1 8" 2>&1 | grep -v amdgpu.ids
    done
  done
done
