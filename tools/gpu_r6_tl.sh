#!/bin/bash
# rocprofv3 kernel trace + timeline of one configuration:  CFG=library TAG=r6d tools/gpu_r6_tl.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
CFG=${CFG:-library}; TAG=${TAG:-r6tl}
mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
ROOTD=$(pwd); O=$ROOTD/gpurun_out/$TAG
B="--config $CFG --steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line ${BENCH_ARGS:-}"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof2 -o trace -- python $ROOTD/bench.py $B > $O/prof_bench.log 2>&1 )
python tools/prof_summary.py $O/prof2 13 > $O/${CFG}_kernel_stats_three_streams.txt 2>> $O/err.log
python tools/prof_timeline.py $O/prof2 3 > $O/${CFG}_timeline_three_streams.txt 2>> $O/err.log
rm -rf $O/prof2
head -8 $O/${CFG}_timeline_three_streams.txt | cut -c1-160
