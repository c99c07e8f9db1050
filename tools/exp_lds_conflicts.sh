#!/bin/bash
# Which phase of a step launch makes the LDS bank conflicts?  The diagnostic build (tools/libmgrid_exp.so) runs the step
# launch of configs 3 and 4 with one phase switched off at a time (MGRID_DBG bits: 1 = no observation, 2 = no copy-out of
# images, 8 = agents not drawn, 32 = no sub-tile passes at all) and with actions that do nothing ("done"); rocprofv3 counts
# conflict cycles and active LDS cycles per launch.  Usage: tools/exp_lds_conflicts.sh OUTDIR   (one counter pass per variant)
O=${1:-gpurun_out/lds}
mkdir -p "$O"
export TMPDIR=/tmp MGRID_TUNING=1 MGRID_LIB=tools/libmgrid_exp.so
for cases in 1 2; do
  for v in "0 random" "0 done" "0 turn" "1 random" "2 random" "3 random" "8 random" "32 random" "32 done"; do
    set -- $v
    export MGRID_DBG=$1 CASES=$cases VARIANTS=$2
    d="$O/c${cases}_dbg$1_$2"
    timeout -s KILL 100 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_LDS -d "$d" -o pmc --output-format csv -- python3 tools/exp_phase.py > "$d.out" 2> "$d.err" || { echo "failed: $d"; exit 1; }
    echo "== case $cases dbg $1 actions $2"
    python3 tools/pmc_summary.py "$d"
  done
done
