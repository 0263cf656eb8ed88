#!/bin/bash
# Round-4 evidence set in ONE gpurun call: full GPU test suite, bench lines (default + other configs), rocprofv3 kernel
# traces (single stream / three streams), PMC passes (HBM traffic, SQ, MFMA utilisation incl. its calibration).
#   SKIP_TESTS=1 tools/gpu_round4.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOTD=$(pwd); O=$ROOTD/gpurun_out; T=r04
LOG=$O/${T}_round.log; : > $LOG
run() { echo "=== $* ===" | tee -a $LOG; local t0=$SECONDS; timeout "${TMO:-600}" "$@" >> $LOG 2>&1; echo "--- rc=$? ($((SECONDS-t0)) s) ---" | tee -a $LOG; }
nproc >> $LOG; lscpu | grep -E "Model name|^CPU\(s\)" >> $LOG
run python __graft_entry__.py build
P="--steps 3 --warmup 2 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
PRE="env LD_PRELOAD=$ROOTD/deep-image-prior_amd/lib/libdip_hip.so DIP_TWO_STREAMS=0"
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && TMO=600 run rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_$ctr -o pmc -- $PRE python $ROOTD/bench.py $P )
done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/${T}_pmc_traffic.json 2>> $LOG
python tools/pmc_summary.py $O/pmc_FETCH_SIZE > $O/${T}_rocprofv3_pmc_FETCH_SIZE.txt 2>> $LOG
python tools/pmc_summary.py $O/pmc_WRITE_SIZE > $O/${T}_rocprofv3_pmc_WRITE_SIZE.txt 2>> $LOG
( cd /tmp && TMO=600 run rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/pmc_SQ -o pmc -- $PRE python $ROOTD/bench.py $P )
python tools/pmc_sq.py $O/pmc_SQ > $O/${T}_rocprofv3_pmc_SQ.txt 2>> $LOG
( cd /tmp && TMO=300 run rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_cal/a -o pmc -- $ROOTD/tools/ubench/bin/mfma_peak )
( cd /tmp && TMO=300 run rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_cal/b -o pmc -- $ROOTD/tools/ubench/bin/bf16x9 )
( cd /tmp && TMO=600 run rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_MFMA -o pmc -- $PRE python $ROOTD/bench.py $P )
python tools/pmc_mfma.py $O/pmc_cal $O/pmc_MFMA $O/${T}_pmc_mfma.json > $O/${T}_rocprofv3_pmc_MFMA.txt 2>> $LOG
# the bench lines below carry these counter figures (bench.py reads profiles/r04_pmc_*.json): same call, same box
cp $O/${T}_pmc_traffic.json $O/${T}_pmc_mfma.json profiles/ 2>> $LOG
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  TMO=600 run python __graft_entry__.py smoke
  TMO=3000 run python -m pytest tests -q -m gpu --no-header -p no:cacheprovider
fi
TMO=900 run python bench.py --steps 100 --warmup 10 --dump-ops $O/${T}_ops.json
grep '^{"metric"' $LOG | tail -1 > $O/${T}_bench_line.json
for cfg in sr kate library snail; do
  TMO=600 run python bench.py --config $cfg --steps 60 --warmup 10 --no-cpu-baseline
  grep '^{"metric"' $LOG | tail -1 > $O/${T}_bench_$cfg.json
done
for cfg in snail library; do
  TMO=600 run env DIP_TWO_STREAMS=0 python bench.py --config $cfg --instances 8 --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-eager-line
  grep '^{"metric"' $LOG | tail -1 > $O/${T}_bench_${cfg}_x8.json
done
B="--steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
( cd /tmp && TMO=900 run rocprofv3 --kernel-trace --stats -d $O/prof1 -o trace -- env DIP_TWO_STREAMS=0 python $ROOTD/bench.py $B )
python tools/prof_summary.py $O/prof1 13 > $O/${T}_rocprofv3_kernel_stats_single_stream.txt 2>> $LOG
python tools/prof_timeline.py $O/prof1 3 > $O/${T}_timeline_single_stream.txt 2>> $LOG
( cd /tmp && TMO=900 run rocprofv3 --kernel-trace --stats -d $O/prof2 -o trace -- python $ROOTD/bench.py $B )
python tools/prof_summary.py $O/prof2 13 > $O/${T}_rocprofv3_kernel_stats_three_streams.txt 2>> $LOG
python tools/prof_timeline.py $O/prof2 3 > $O/${T}_timeline_three_streams.txt 2>> $LOG
tools/ubench/bin/bf16x9 > $O/${T}_ubench_bf16x9.txt 2>&1
rm -rf $O/prof1 $O/prof2 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ $O/pmc_cal $O/pmc_MFMA
grep -E "passed|failed|error|rc=|^FAILED|^ERROR" $LOG | tail -40
