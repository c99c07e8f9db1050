#!/bin/bash
# Profiling passes of one round on the GPU box: rocprofv3 kernel-trace statistics and PMC passes (each counter group in
# a run of its own) of bench.py workloads.  Usage: tools/profile_round.sh OUTDIR [quick]
# Every pass runs the program itself after "--" and has its own time limit; the script stops at the first failure.
set -e
O=${1:-gpurun_out/prof}
mkdir -p "$O"
export TMPDIR=/tmp
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
SQ2="SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
kt() {  # name, bench arguments
    local name=$1; shift
    timeout -s KILL 170 rocprofv3 --kernel-trace --stats -d "$O/kt_$name" -o kt --output-format csv -- python3 bench.py "$@" --no-cpu --no-extra > "$O/kt_$name.json" 2> "$O/kt_$name.err"
    echo "kt $name done"
}
pmc() {  # name, counters, bench arguments
    local name=$1 ctr=$2; shift 2
    timeout -s KILL 170 rocprofv3 --pmc $ctr -d "$O/pmc_$name" -o pmc --output-format csv -- python3 bench.py "$@" --no-cpu --no-extra > "$O/pmc_$name.json" 2> "$O/pmc_$name.err"
    echo "pmc $name done"
}
workload() {  # name, sq?, bench arguments
    local name=$1 sq=$2; shift 2
    kt "$name" "$@" --steps 400 --warmup 40
    pmc "${name}_fetch" FETCH_SIZE "$@" --steps 20 --warmup 5 --repeats 1
    pmc "${name}_write" WRITE_SIZE "$@" --steps 20 --warmup 5 --repeats 1
    if [ "$sq" = sq ]; then
        pmc "${name}_sq1" "$SQ1" "$@" --steps 20 --warmup 5 --repeats 1
        pmc "${name}_sq2" "$SQ2" "$@" --steps 20 --warmup 5 --repeats 1
    fi
}
workload cfg5_131072 sq --config 5
workload cfg5_1048576 nosq --config 5 --envs 1048576
workload cfg4_65536 sq --config 4
workload cfg4_262144 nosq --config 4 --envs 262144
workload cfg3_65536 sq --config 3
workload cfg2_4096 sq --config 2
# fused launches (mg_step_n, 8 steps per launch) and the no-overlap variant: kernel-trace only
kt cfg3_65536_T8 --config 3 --fused 8 --steps 384 --warmup 64
kt cfg4_65536_T8 --config 4 --fused 8 --steps 384 --warmup 64
kt cfg5_131072_T8 --config 5 --fused 8 --steps 128 --warmup 32
kt cfg5_131072_nooverlap --config 5 --no-overlap --steps 400 --warmup 40
echo "all passes done"
