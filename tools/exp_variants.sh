#!/bin/bash
# Compare tuning builds of the library (tools/libmgrid_<name>.so; "product" = the in-tree library) on chosen workloads:
#   tools/exp_variants.sh "vA vB product" "3 1" "5 8" "5 1 --no-overlap" ...      (config, steps per launch, extra bench flags)
set -e
libs=$1; shift
for lib in $libs; do
  for w in "$@"; do
    cfg=${w%% *}; rest=${w#* }; T=${rest%% *}; extra=""; [ "$rest" != "$T" ] && extra=${rest#* }
    if [ "$lib" = product ]; then unset MGRID_LIB; else export MGRID_LIB=$PWD/tools/libmgrid_$lib.so; fi
    timeout -k 10 120 python3 bench.py --config $cfg --fused $T $extra --steps 384 --warmup 64 --no-cpu --no-extra \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg$cfg T=$T $extra', 'us/step %.2f' % (d['roofline']['launch_us']/$T), 'frac %.3f' % d['roofline']['frac'])"
  done
done
