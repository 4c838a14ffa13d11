#!/bin/bash
# Runs ON the GPU box (gpurun): for `bench.py --workload c4 --steps 2 --warmup 8` the rocprofv3 kernel statistics and four separate --pmc
# passes (SQ, SQ/LDS, FETCH_SIZE, WRITE_SIZE).  (The default bench line and C4's steady-state line come from scripts/refresh_profiles.sh.)
# Outputs land in gpurun_out/final/; scripts/c4_collect.py turns them into profiles/<round>_c4_counters.json and <round>_c4_kernel_stats.csv.
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
rm -rf $OUT && mkdir -p $OUT
( while sleep 45; do date >> $OUT/heartbeat; done ) &
HB=$!
trap "kill $HB" EXIT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_stats -o run -- $B --workload c4 --steps 2 --warmup 8 --no-cpu-baseline > $OUT/c4_stats.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $OUT/c4_sq -o run -- $B --workload c4 --steps 2 --warmup 8 --no-cpu-baseline > $OUT/c4_sq.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/c4_sq2 -o run -- $B --workload c4 --steps 2 --warmup 8 --no-cpu-baseline > $OUT/c4_sq2.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c4_fetch -o run -- $B --workload c4 --steps 2 --warmup 8 --no-cpu-baseline > $OUT/c4_fetch.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c4_write -o run -- $B --workload c4 --steps 2 --warmup 8 --no-cpu-baseline > $OUT/c4_write.log 2>&1 || exit 1
echo "[c4_profiles] done"
