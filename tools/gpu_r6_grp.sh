#!/bin/bash
# rocprofv3 kernel trace of the grouped 'library' iteration (8 fits, one launch list), eager, default streams
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6g; export TMPDIR=/tmp
ROOTD=$(pwd); O=$ROOTD/gpurun_out/r6g
B="--config library --instances 8 --group native --mode eager --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-eager-line"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/profg -o trace -- python $ROOTD/bench.py $B > $O/prof.log 2>&1 )
python tools/prof_summary.py $O/profg 8 > $O/library_x8_kernel_stats.txt 2>> $O/err.log
python tools/prof_timeline.py $O/profg 3 > $O/library_x8_timeline.txt 2>> $O/err.log
rm -rf $O/profg
head -45 $O/library_x8_kernel_stats.txt | cut -c1-170; head -8 $O/library_x8_timeline.txt | cut -c1-150
