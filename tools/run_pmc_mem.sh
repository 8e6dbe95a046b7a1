#!/bin/bash
# usage: run_pmc_mem.sh <outdir> <kind> [H Cin Cout B]   -- memory-path counter passes for one conv shape
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="python3 $GRAFT_REPO_ROOT/tools/pmc_conv.py $*"
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- $P > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; exit 1; }; }
run m1 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM
run m2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum SQ_WAIT_ANY SQ_INSTS_VMEM_RD
run m3 TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TCP_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum
run m4 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum
run m5 TCC_READ_REQ_LATENCY_sum TCC_READ_REQ_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum
echo ok
