#!/bin/bash
# same-box A/B of library builds: scripts/ab_bench.sh <rounds> <lib> <lib> ...   (run ON the GPU box; box-to-box spread is +-4 %)
R=${GRAFT_REPO_ROOT:-/root/repo}
rounds=$1; shift
for i in $(seq $rounds); do
  for lib in "$@"; do
    FBA_LIB=$R/fba_pomdp_amd/$lib timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline ${AB_ARGS} > /tmp/ab.json 2>/tmp/ab.err
    python3 -c "
import json,sys
d=json.load(open('/tmp/ab.json')); print(sys.argv[1], round(d['ms_per_step'],2), round(d['search_kernel']['avg_ms'],2), round(d['roofline']['avg_ms'],2))" $lib
  done
done
