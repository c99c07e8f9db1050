#!/bin/bash
# Further SQ counters of the step launch (instruction fetch, vector-memory issue, FIFO-full stalls, levels) for configs
# 5, 3 and 4; one rocprofv3 pass per counter group.  Usage: tools/exp_sq_more.sh OUTDIR
O=${1:-gpurun_out/sqmore}
mkdir -p "$O"
export TMPDIR=/tmp
G1="SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL"
G2="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAVE_CYCLES SQ_WAVES"
G3="SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES SQ_CYCLES"
for cfg in 5 3 4; do
  k=0
  for g in "$G1" "$G2" "$G3"; do
    k=$((k+1))
    d="$O/cfg${cfg}_g$k"
    timeout -s KILL 100 rocprofv3 --pmc $g -d "$d" -o pmc --output-format csv -- python3 bench.py --config $cfg --steps 20 --warmup 5 --repeats 1 --no-cpu --no-extra > "$d.json" 2> "$d.err" || { echo "failed $d"; exit 1; }
    echo "== cfg $cfg group $k"
    python3 tools/pmc_summary.py "$d"
  done
done
