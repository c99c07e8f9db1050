#!/bin/bash
# A/B: header blocks written back without first draining the launch's stores (tools/libmgrid_loopwait.so: the loop as it
# was plus an explicit wait at the end of the rare restart path; tools/libmgrid_unrolled.so: also straight-line stores)
# against the product build.  Two rounds, so that drift of the box shows.
for round in 1 2; do
for l in "" tools/libmgrid_loopwait.so tools/libmgrid_unrolled.so; do
  echo "== round $round ${l:-product}"
  if [ -z "$l" ]; then unset MGRID_LIB; else export MGRID_LIB=$l; fi
  bash tools/exp_quick.sh "3 1" "4 1" "5 1" "2 1" 2>&1 | grep -v amdgpu.ids
done
done
