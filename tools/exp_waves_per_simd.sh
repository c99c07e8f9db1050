#!/bin/bash
# Step time against the number of wavefronts per SIMD (1024 SIMDs; config 3 puts 32 environments, config 4 16 x 2, in
# a wavefront): is a launch bound by one wavefront's chain (flat) or by what a SIMD issues (linear)?
for cfg in 3 4; do
  for envs in 16384 32768 49152 65536 98304 131072 196608; do
    timeout -k 10 120 python3 bench.py --config $cfg --envs $envs --steps 384 --warmup 64 --no-cpu --no-extra \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg$cfg envs $envs', 'waves per SIMD %.2f' % ($envs/32/1024), 'us/step %.2f' % d['roofline']['launch_us'])"
  done
done
