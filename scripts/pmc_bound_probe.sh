#!/bin/bash
# Runs ON the GPU box: counter passes that say what the bench's search kernel waits for (wave-level waits, the texture-address / L1 pipeline, the L1 TLB).
# Output: gpurun_out/bound/<pass>/run_counter_collection.csv; scripts/pmc_bound_summary.py prints the per-kernel sums.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/bound
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 4 --warmup 0 --no-cpu-baseline"
run() { timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o run -- $B > $OUT/$1.log 2>&1 || { echo "[pass $1 failed]"; grep -m1 "error code" $OUT/$1.log || true; }; }
# (a pass holds what one block's counter registers can: four TCP, two TA)
run sq   "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES"
run sq2  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SALU"
run ta1  "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY"
run ta2  "TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
run tlb1 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
run tlb2 "TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
run tlb3 "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_LFIFO_FULL_sum"
run tcp3 "TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum"
echo done
