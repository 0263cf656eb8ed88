#!/bin/bash
# One gpurun call: per-group GPU tests (each group in its own process so a GPU fault in one group
# does not hide the others), smoke, bench and rocprof kernel traces.  Logs -> gpurun_out/.
#   SKIP_TESTS=1 SKIP_BENCH=1 DO_PROF=1 DO_PMC=1 EXTRA_BENCH="kate library snail sr" TESTS="..." tools/gpu_round.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOTD=$(pwd)
LOG=$ROOTD/gpurun_out/round.log
: > $LOG
run() { echo "=== $* ===" | tee -a $LOG; local t0=$SECONDS; timeout "${TMO:-600}" "$@" >> $LOG 2>&1; echo "--- rc=$? ($((SECONDS-t0)) s) ---" | tee -a $LOG; }
nproc >> $LOG; lscpu | grep -E "Model name|^CPU\(s\)" >> $LOG
# the pool has two speed classes of boxes (~128 and ~140 it/s on the default config): record what this one reports
(rocm-smi --showperflevel --showclocks --showpower --showmaxpower 2>/dev/null | grep -vE "^=|^$" | head -40) >> $LOG
run python __graft_entry__.py build
if [ "${SKIP_TESTS:-0}" != "1" ]; then
if [ -n "${TESTS:-}" ]; then
  TMO=1500 run python -m pytest $TESTS -q -m gpu --no-header -p no:cacheprovider -s
else
TMO=600 run python __graft_entry__.py smoke
TMO=1500 run python -m pytest tests/test_fullsize_gpu.py -q -m gpu --no-header -p no:cacheprovider -s
TMO=900 run python -m pytest tests/test_closure_gpu.py tests/test_monitor_gpu.py -q -m gpu --no-header -p no:cacheprovider -s
if [ "${EQ:-1}" = "1" ]; then
TMO=2400 run python -m pytest tests/test_net_gpu.py -q -m gpu --no-header -p no:cacheprovider -s
else
TMO=1500 run python -m pytest tests/test_net_gpu.py -q -m gpu -k "not end_quality" --no-header -p no:cacheprovider -s
fi
TMO=900 run python -m pytest tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider
fi
fi
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  TMO=900 run python bench.py --steps 100 --warmup 10 --dump-ops gpurun_out/ops.json
  grep '^{"metric"' $LOG | tail -1 > gpurun_out/bench.json
  for cfg in ${EXTRA_BENCH:-}; do
    TMO=600 run python bench.py --config $cfg --steps 60 --warmup 10 --no-cpu-baseline --dump-ops gpurun_out/ops_$cfg.json
    grep '^{"metric"' $LOG | tail -1 > gpurun_out/bench_$cfg.json
  done
  if [ -n "${BENCH_INSTANCES:-}" ]; then
    for cfg in snail library; do
      TMO=600 run env DIP_TWO_STREAMS=0 python bench.py --config $cfg --instances $BENCH_INSTANCES --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-eager-line
      grep '^{"metric"' $LOG | tail -1 > gpurun_out/bench_${cfg}_x$BENCH_INSTANCES.json
    done
  fi
fi
if [ "${DO_PROF:-0}" = "1" ]; then
  ( cd /tmp && TMO=900 run rocprofv3 --kernel-trace --stats -d $ROOTD/gpurun_out/prof1 -o trace -- env DIP_TWO_STREAMS=0 python $ROOTD/bench.py --steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line )
  python tools/prof_summary.py gpurun_out/prof1 13 > gpurun_out/prof1_summary.txt 2>> $LOG
  if [ "${DO_PROF2:-0}" = "1" ]; then
  ( cd /tmp && TMO=900 run rocprofv3 --kernel-trace --stats -d $ROOTD/gpurun_out/prof2 -o trace -- python $ROOTD/bench.py --steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line )
  python tools/prof_summary.py gpurun_out/prof2 13 > gpurun_out/prof2_summary.txt 2>> $LOG
  python tools/prof_timeline.py gpurun_out/prof2 3 > gpurun_out/prof2_timeline.txt 2>> $LOG
  fi
fi
if [ "${DO_PMC:-0}" = "1" ]; then
  # counters in their own passes (no trace domains besides --kernel-trace); the library is preloaded
  # because rocprofv3's counter service crashes on code objects that are dlopen()ed after start-up
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && TMO=600 run rocprofv3 --kernel-trace --pmc $ctr -d $ROOTD/gpurun_out/pmc_$ctr -o pmc -- env LD_PRELOAD=$ROOTD/deep-image-prior_amd/lib/libdip_hip.so python $ROOTD/bench.py --steps 3 --warmup 2 --mode eager --no-cpu-baseline --no-roofline --no-eager-line )
  done
  python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_traffic.json 2>> $LOG
  python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE > gpurun_out/pmc_FETCH_SIZE_summary.txt 2>> $LOG
  python tools/pmc_summary.py gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_WRITE_SIZE_summary.txt 2>> $LOG
  rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
fi
rm -f gpurun_out/prof1/*.db gpurun_out/prof2/*.db
grep -E "passed|failed|error|rc=|^FAILED|^ERROR" $LOG | tail -60
