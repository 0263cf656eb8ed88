#!/bin/bash
# One gpurun call: per-group GPU tests (each group in its own process so a GPU fault in one group
# does not hide the others), smoke, a short bench and rocprof kernel traces.  Logs -> gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOTD=$(pwd)
LOG=$ROOTD/gpurun_out/round.log
: > $LOG
run() { echo "=== $* ===" | tee -a $LOG; timeout "${TMO:-600}" "$@" >> $LOG 2>&1; echo "--- rc=$? ---" | tee -a $LOG; }
nproc >> $LOG; lscpu | grep -E "Model name|^CPU\(s\)" >> $LOG
run python __graft_entry__.py build
if [ "${SKIP_TESTS:-0}" != "1" ]; then
for grp in "conv_forward" "conv_dgrad" "conv_wgrad" "mfma or bn_forward or upcat or avgpool or layout or adam or noise or lanczos"; do
  TMO=900 run python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "$grp" --no-header -p no:cacheprovider
done
TMO=1200 run python -m pytest tests/test_net_gpu.py tests/test_monitor_gpu.py -q -m gpu --no-header -p no:cacheprovider -s
TMO=600 run python __graft_entry__.py smoke
fi
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  TMO=900 run python bench.py --steps 60 --warmup 10 --dump-ops gpurun_out/ops.json
  grep '^{"metric"' $LOG | tail -1 > gpurun_out/bench.json
fi
if [ "${DO_PROF:-0}" = "1" ]; then
  ( cd /tmp && TMO=900 run rocprofv3 --kernel-trace --stats -d $ROOTD/gpurun_out/prof1 -o trace -- env DIP_TWO_STREAMS=0 python $ROOTD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline )
  ( cd /tmp && TMO=900 run rocprofv3 --kernel-trace --stats -d $ROOTD/gpurun_out/prof2 -o trace -- python $ROOTD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline )
fi
if [ "${DO_PMC:-0}" = "1" ]; then
  # counters in their own passes (no trace domains besides --kernel-trace); the library is preloaded
  # because rocprofv3's counter service crashes on code objects that are dlopen()ed after start-up
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && TMO=600 run rocprofv3 --kernel-trace --pmc $ctr -d $ROOTD/gpurun_out/pmc_$ctr -o pmc -- env LD_PRELOAD=$ROOTD/deep-image-prior_amd/lib/libdip_hip.so python $ROOTD/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline )
  done
  python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_traffic.json 2>> $LOG
fi
grep -E "passed|failed|error|rc=" $LOG | tail -40
