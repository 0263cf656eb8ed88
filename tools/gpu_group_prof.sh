#!/bin/bash
# rocprofv3 kernel traces of the grouped iteration (8 fits, one launch list) of the small configurations, and of the
# one-graph-per-fit form next to it.   gpurun --timeout 300 -- 'bash tools/gpu_group_prof.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOTD=$(pwd); O=$ROOTD/gpurun_out; T=r04g
LOG=$O/${T}_prof.log; : > $LOG
run() { echo "=== $* ===" | tee -a $LOG; local t0=$SECONDS; timeout "${TMO:-120}" "$@" >> $LOG 2>&1; echo "--- rc=$? ($((SECONDS-t0)) s, t=$SECONDS) ---" | tee -a $LOG; }
B="--instances 8 --group native --mode eager --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-eager-line"
for cfg in snail library; do
  ( cd /tmp && TMO=120 run rocprofv3 --kernel-trace --stats -d $O/profg_$cfg -o trace -- env DIP_TWO_STREAMS=0 python $ROOTD/bench.py --config $cfg $B )
  python tools/prof_summary.py $O/profg_$cfg 8 > $O/${T}_rocprofv3_kernel_stats_${cfg}_x8_grouped.txt 2>> $LOG
  python tools/prof_timeline.py $O/profg_$cfg 3 > $O/${T}_timeline_${cfg}_x8_grouped.txt 2>> $LOG
  rm -rf $O/profg_$cfg
done
( cd /tmp && TMO=120 run rocprofv3 --kernel-trace --stats -d $O/profs -o trace -- env DIP_TWO_STREAMS=0 python $ROOTD/bench.py --config library --mode eager --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-eager-line )
python tools/prof_summary.py $O/profs 8 > $O/${T}_rocprofv3_kernel_stats_library_solo.txt 2>> $LOG
rm -rf $O/profs
head -40 $O/${T}_rocprofv3_kernel_stats_library_x8_grouped.txt
