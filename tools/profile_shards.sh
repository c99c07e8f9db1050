#!/bin/bash
# rocprofv3 passes of the stream-shard mode (bench.py --stream-shards 2) on the GPU box: kernel trace with statistics,
# then FETCH_SIZE and WRITE_SIZE, every pass a run of its own with the program itself after "--".
# Usage: tools/profile_shards.sh OUTDIR
set -e
O=${1:-gpurun_out/prof_shards}
mkdir -p "$O"
export TMPDIR=/tmp
one() {  # name, bench arguments
    local name=$1; shift
    timeout -s KILL 170 rocprofv3 --kernel-trace --stats -d "$O/kt_$name" -o kt --output-format csv -- python3 bench.py "$@" --no-cpu --no-extra --steps 400 --warmup 40 > "$O/kt_$name.json" 2> "$O/kt_$name.err"
    echo "kt $name done"
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -s KILL 170 rocprofv3 --pmc $c -d "$O/pmc_${name}_$c" -o pmc --output-format csv -- python3 bench.py "$@" --no-cpu --no-extra --steps 20 --warmup 5 --repeats 1 > "$O/pmc_${name}_$c.json" 2> "$O/pmc_${name}_$c.err"
        echo "pmc $name $c done"
    done
}
one cfg5_131072_s1 --config 5
one cfg5_131072_s2 --config 5 --stream-shards 2
one cfg4_262144_s2 --config 4 --envs 262144 --stream-shards 2
one cfg4_65536_s2 --config 4 --stream-shards 2
python3 tools/prof_shards_summary.py "$O" > "$O/summary.json"
echo "all passes done"
