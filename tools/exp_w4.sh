#!/bin/bash
# A/B: non-see-through step kernels held to 128 vector registers (four wavefronts per SIMD, a few spilled registers)
# against the product build, with tiles of one and two sub-tiles.  tools/libmgrid_w4.so: -DMG_ONLY_VIEW7 -DMG_WAVES_SINGLE=4
export MGRID_TUNING=1
for lib in "" tools/libmgrid_w4.so; do
  for tile in "" "16,1"; do
    echo "== lib=${lib:-product} tile=${tile:-auto}"
    export MGRID_LIB=$lib MGRID_TILE=$tile
    [ -z "$lib" ] && unset MGRID_LIB
    [ -z "$tile" ] && unset MGRID_TILE
    bash tools/exp_quick.sh "4 1" "3 1" 2>&1 | grep -v amdgpu.ids
    timeout -k 10 120 python3 bench.py --config 4 --envs 262144 --steps 384 --warmup 64 --no-cpu --no-extra | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4 262144', 'us/step %.2f' % d['roofline']['launch_us'], 'frac %.3f' % d['roofline']['frac'])"
  done
done
