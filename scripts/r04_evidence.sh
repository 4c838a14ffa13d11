#!/bin/bash
# Runs ON the GPU box: round-4 evidence that needs no new kernel -- (1) 64- vs 128-byte lines (randline pair64 / line128 / far2x64, with
# FETCH_SIZE and TCC request counters), (2) C3's rejection kernel traffic (FETCH_SIZE / WRITE_SIZE passes), (3) C4 steady state (warm-up 20).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_evidence
mkdir -p $OUT
( while sleep 45; do date >> $OUT/heartbeat; done ) &
HB=$!
trap "kill $HB" EXIT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
if [ "${SKIP_RANDLINE:-0}" != 1 ]; then
timeout -k 10 300 $R/scripts/micro/randline 32 2048 > $OUT/randline.jsonl 2> $OUT/randline.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/rl_fetch -o run -- $R/scripts/micro/randline 32 1024 > $OUT/rl_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/rl_l2 -o run -- $R/scripts/micro/randline 32 1024 > $OUT/rl_l2.log 2>&1 || exit 1
echo "[r04] randline done"
fi
if [ "${SKIP_C3:-0}" != 1 ]; then
S="--workload c3 --steps 4 --warmup 0 --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c3_fetch -o run -- $B $S > $OUT/c3_fetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c3_write -o run -- $B $S > $OUT/c3_write.log 2>&1 || exit 1
echo "[r04] c3 pmc done"
fi
if [ "${SKIP_C4:-0}" != 1 ]; then
timeout -k 10 1000 $B --workload c4 --warmup ${C4_WARMUP:-20} --steps ${C4_STEPS:-10} --no-cpu-baseline > $OUT/c4_steady.json 2> $OUT/c4_steady.err || exit 1
cat $OUT/c4_steady.json
fi
