#!/bin/bash
# Runs ON the GPU box (gpurun): the default bench line, its rocprofv3 kernel statistics, and the two
# PMC passes (FETCH_SIZE, WRITE_SIZE) of the same command.  Outputs land in gpurun_out/refresh/.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/refresh
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $R/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
timeout -k 10 300 python3 $R/bench.py --belief importance_sampling --no-cpu-baseline > $OUT/bench_importance.json 2> $OUT/bench_importance.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/l2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/sq.log 2>&1
find $OUT -name "*.csv" | head -20
cat $OUT/bench_default.json
