set -e
O=gpurun_out/prof_shards_t8
mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 170 rocprofv3 --kernel-trace --stats -d $O/kt_cfg5_131072_s2_T8 -o kt --output-format csv -- python3 bench.py --config 5 --stream-shards 2 --fused 8 --no-cpu --no-extra --steps 384 --warmup 64 > $O/kt_cfg5_131072_s2_T8.json 2> $O/kt_cfg5_131072_s2_T8.err
python3 tools/prof_shards_summary.py $O > $O/summary.json
cat $O/summary.json
