#!/bin/bash
# A/B of a tuning build of the library (MGRID_LIB) against the product build on chosen workloads, on one box:
#   tools/exp_lib_ab.sh tools/libmgrid_o3.so "3 1" "4 1" "5 1" "5 8"     (config, steps per launch)
lib=$1; shift
for l in "" "$lib"; do
  echo "== ${l:-product}"
  if [ -z "$l" ]; then unset MGRID_LIB; else export MGRID_LIB=$l; fi
  bash tools/exp_quick.sh "$@" 2>&1 | grep -v amdgpu.ids
done
