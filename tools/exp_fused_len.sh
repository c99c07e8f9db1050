#!/bin/bash
# Fused launch length against batch size on config 5: is the slowdown from 32 steps per launch on a matter of how many
# bytes one launch's outputs span, or of how far the wavefronts of one launch drift apart?
for envs in 65536 131072 262144; do
  for t in 8 16 32; do
    timeout -k 10 120 python3 bench.py --config 5 --envs $envs --fused $t --steps 192 --warmup 64 --no-cpu --no-extra \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 envs $envs T=$t', 'us/step %.2f' % (d['roofline']['launch_us']/$t), 'per 131072 envs %.2f' % (d['roofline']['launch_us']/$t*131072/$envs), 'output GB per launch %.2f' % ($envs*8*147*$t/1e9))"
  done
done
