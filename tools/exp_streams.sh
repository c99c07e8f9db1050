#!/bin/bash
# same-box sweep of the stream-shard experiment
set -e
export MGRID_TUNING=1  # the library reads its tuning switches only with this opt-in
T="timeout -k 10 120"
$T python -u tools/exp_streams.py 2 3 4 5 --graph --shards 1 2 4 8
MGRID_TILE=32,1 $T python -u tools/exp_streams.py 3 --graph --shards 2 4
MGRID_TILE=16,2 $T python -u tools/exp_streams.py 4 --graph --shards 2 4
$T python -u tools/exp_streams.py 4 --envs 262144 --shards 1 2 4 --steps 400 --graph
