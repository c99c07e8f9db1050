#!/bin/bash
# Quick timings of chosen workloads with the product library: tools/exp_quick.sh "5 1" "5 8" "5n 1" ...  (config, steps per launch;
# an "n" after the config number = allow_agent_overlap=False is not available from the command line: see bench.py other_configs)
set -e
for w in "$@"; do
  set -- $w
  timeout -k 10 120 python3 bench.py --config $1 --fused $2 --steps 384 --warmup 64 --no-cpu --no-extra \
    | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg$1 T=$2', 'us/step %.2f' % (d['roofline']['launch_us']/$2), 'frac %.3f' % d['roofline']['frac'])"
done
