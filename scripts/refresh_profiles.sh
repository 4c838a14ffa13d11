#!/bin/bash
# Runs ON the GPU box (gpurun): the default bench line, its rocprofv3 kernel statistics, the PMC passes (FETCH_SIZE,
# WRITE_SIZE, L2, SQ) of the same command, the two traffic passes for the importance filter of the bench workload, the C5
# shape (line, kernel statistics, traffic) and the random-sector micro-benchmark the search kernel's bound is quoted
# against.  Outputs land in gpurun_out/refresh/; scripts/collect_profiles.py turns them into profiles/<round>_*.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/refresh
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
S="--steps 4 --warmup 0 --no-cpu-baseline"    # (warm-up 0: every search launch of the run is inside the bench line's own step count)
timeout -k 10 400 $B > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $B --no-cpu-baseline > $OUT/stats.log 2>&1
timeout -k 10 300 $B --belief importance_sampling --no-cpu-baseline > $OUT/bench_importance.json 2> $OUT/bench_importance.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- $B $S > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- $B $S > $OUT/write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -o run -- $B $S > $OUT/l2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -o run -- $B $S > $OUT/sq.log 2>&1
echo "[refresh] bench passes done"
# the importance filter of the bench workload: traffic passes
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/is_fetch -o run -- $B --belief importance_sampling $S > $OUT/is_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/is_write -o run -- $B --belief importance_sampling $S > $OUT/is_write.log 2>&1
echo "[refresh] importance passes done"
# C5 shape: one belief of 10^6 collision-avoidance particles
timeout -k 10 300 python3 $R/scripts/bench_c5.py 1000000 1 10 > $OUT/c5.json 2> $OUT/c5.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -o run -- python3 $R/scripts/bench_c5.py 1000000 1 10 > $OUT/c5_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c5_fetch -o run -- python3 $R/scripts/bench_c5.py 1000000 1 10 > $OUT/c5_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c5_write -o run -- python3 $R/scripts/bench_c5.py 1000000 1 10 > $OUT/c5_write.log 2>&1
echo "[refresh] c5 done"
# what the memory system gives for one random 64-byte record per lane (the search kernel's access shape), with its counters
if [ -x $R/scripts/micro/randline ]; then
  timeout -k 10 300 $R/scripts/micro/randline 32 2048 > $OUT/randline.jsonl
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/rl_fetch -o run -- $R/scripts/micro/randline 32 1024 > $OUT/rl_fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/rl_write -o run -- $R/scripts/micro/randline 32 1024 > $OUT/rl_write.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/rl_l2 -o run -- $R/scripts/micro/randline 32 1024 > $OUT/rl_l2.log 2>&1
fi
echo "[refresh] randline done"
# the other BASELINE configs through the same bench.py (c4 apart: its steady state needs a warm-up past one whole first episode)
for wl in c1 c3 c5; do
  timeout -k 10 400 $B --workload $wl --steps 4 --warmup 1 > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || echo "[refresh] $wl failed"
done
( while sleep 60; do echo "[refresh] c4 running"; done ) &
HB=$!
timeout -k 10 900 $B --workload c4 --steps 10 --warmup 20 > $OUT/bench_c4.json 2> $OUT/bench_c4.err || echo "[refresh] c4 failed"
kill $HB
# C3's rejection kernel: traffic passes
S3="--workload c3 --steps 4 --warmup 0 --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c3_fetch -o run -- $B $S3 > $OUT/c3_fetch.log 2>&1 || echo "[refresh] c3 fetch failed"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c3_write -o run -- $B $S3 > $OUT/c3_write.log 2>&1 || echo "[refresh] c3 write failed"
cat $OUT/bench_default.json
cat $OUT/c5.json
