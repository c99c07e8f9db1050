#!/bin/bash
# Compare tuning builds of the library (tools/libmgrid_v*.so) on the launch-bound and the large workloads.
set -e
for lib in "$@"; do
  for cfg in 3 4 5; do
    for T in 1 8; do
      MGRID_LIB=$PWD/tools/libmgrid_$lib.so timeout -k 10 120 python3 bench.py --config $cfg --fused $T --steps 384 --warmup 64 --no-cpu --no-extra \
        | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg$cfg T=$T', 'us/step %.2f' % (d['roofline']['launch_us']/$T), 'frac %.3f' % d['roofline']['frac'])"
    done
  done
done
