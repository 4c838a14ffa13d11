set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1
FBA_SEARCH_SORT=0 python scripts/tick_probe.py 262144 8 2>&1 | tee gpurun_out/r1/tick_probe.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r1/gputests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r1/gputests.log
tail -5 gpurun_out/r1/gputests.log
