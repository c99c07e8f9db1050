#!/bin/bash
# Instructions and wavefront clocks of a step launch by phase: the diagnostic build with one phase switched off at a time
# (MGRID_DBG bits as in tools/exp_lds_conflicts.sh), rocprofv3 SQ counters per launch.  Usage: tools/exp_insts_by_phase.sh OUTDIR
O=${1:-gpurun_out/insts}
mkdir -p "$O"
export TMPDIR=/tmp MGRID_TUNING=1 MGRID_LIB=tools/libmgrid_exp.so
for cases in 1 2 0; do
  for v in "0 random" "0 done" "1 random" "2 random" "3 random" "32 random" "32 done"; do
    set -- $v
    export MGRID_DBG=$1 CASES=$cases VARIANTS=$2
    d="$O/c${cases}_dbg$1_$2"
    timeout -s KILL 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES SQ_INSTS_SMEM -d "$d" -o pmc --output-format csv -- python3 tools/exp_phase.py > "$d.out" 2> "$d.err" || { echo "failed: $d"; exit 1; }
    echo "== case $cases dbg $1 actions $2"
    python3 tools/pmc_summary.py "$d"
    grep "us/launch" "$d.out"
  done
done
