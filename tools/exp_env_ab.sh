#!/bin/bash
# A/B of one environment switch of the library on chosen workloads, on one box:
#   tools/exp_env_ab.sh MGRID_STAGE "16 32 64 default" "2 1" "3 1" "5 8" ...   (config, steps per launch, extra flags)
set -e
export MGRID_TUNING=1  # the library reads its tuning switches only with this opt-in
var=$1; vals=$2; shift 2
for val in $vals; do
  for w in "$@"; do
    cfg=${w%% *}; rest=${w#* }; T=${rest%% *}; extra=""; [ "$rest" != "$T" ] && extra=${rest#* }
    if [ "$val" = default ]; then unset $var; else export $var=$val; fi
    timeout -k 10 120 python3 bench.py --config $cfg --fused $T $extra --steps 384 --warmup 64 --no-cpu --no-extra \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$var=$val cfg$cfg T=$T $extra', 'us/step %.2f' % (d['roofline']['launch_us']/$T), 'frac %.3f' % d['roofline']['frac'])"
  done
done
