set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOTD=$(pwd)
LOG=$ROOTD/gpurun_out/round.log
: > $LOG
run() { echo "=== $* ===" | tee -a $LOG; local t0=$SECONDS; timeout "${TMO:-600}" "$@" >> $LOG 2>&1; echo "--- rc=$? ($((SECONDS-t0)) s) ---" | tee -a $LOG; }
run python __graft_entry__.py build
TMO=600 run python -m pytest tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider
TMO=300 run python __graft_entry__.py smoke
TMO=600 run python -m pytest tests/test_closure_gpu.py -q -m gpu --no-header -p no:cacheprovider -k "not single_graph"
TMO=900 run python -m pytest tests/test_net_gpu.py tests/test_monitor_gpu.py -q -m gpu --no-header -p no:cacheprovider -k "not end_quality_default" -s
TMO=900 run python -m pytest tests/test_fullsize_gpu.py -q -m gpu --no-header -p no:cacheprovider -k "appa or default_net"
TMO=600 run python bench.py --steps 100 --warmup 10 --dump-ops gpurun_out/ops.json
grep '^{"metric"' $LOG | tail -1 > gpurun_out/bench.json
TMO=300 run python tools/wgrad_sweep.py
TMO=300 run python bench.py --config snail --steps 100 --warmup 10 --no-cpu-baseline --no-eager-line --dump-ops gpurun_out/ops_snail.json
grep -E "passed|failed|error|rc=|^FAILED|^ERROR" $LOG | tail -40
