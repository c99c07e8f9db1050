#!/bin/bash
# Diagnostic build only (tools/build_exp.sh): staggered tile starts (MGRID_DBG bit 6, see the kernel) against the same
# binary without them, one box, one call.
set -e
export MGRID_TUNING=1  # the library reads its tuning switches only with this opt-in
export MGRID_LIB=tools/libmgrid_exp.so
run() {  # label, dbg, bench arguments
    local label=$1 dbg=$2; shift 2
    MGRID_DBG=$dbg timeout -k 10 120 python3 bench.py "$@" --steps 400 --warmup 40 --no-cpu --no-extra \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label dbg=$dbg', 'us/step %.2f' % d['roofline']['us_per_step'], 'frac %.3f' % d['roofline']['frac'])"
}
for w in "--config 5" "--config 4 --envs 262144" "--config 4" "--config 3"; do
    run "$w" 0 $w
    for shift in 3 10 11 12; do
        for n in 1 2; do
            run "$w shift=$shift n=$n" $((64 + (n << 8) + (shift << 16))) $w
        done
    done
    run "$w" 0 $w
done
