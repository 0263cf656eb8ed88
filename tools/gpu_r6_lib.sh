#!/bin/bash
# round 6, library-net call: new conv_small cases, the per-layer sweep, rocprofv3 timeline of the library iteration
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6b; export TMPDIR=/tmp
ROOTD=$(pwd); O=$ROOTD/gpurun_out/r6b
timeout 600 python -m pytest tests/test_small_gpu.py -x -q -m gpu > $O/test_small.log 2>&1; echo "test_small rc=$?"; tail -3 $O/test_small.log
timeout 300 python tools/thin_sweep.py all > $O/thin_sweep.txt 2>&1; echo "sweep rc=$?"
B="--config library --steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof2 -o trace -- python $ROOTD/bench.py $B > $O/prof_bench.log 2>&1 )
python tools/prof_summary.py $O/prof2 13 > $O/library_kernel_stats_three_streams.txt 2>> $O/err.log
python tools/prof_timeline.py $O/prof2 3 > $O/library_timeline_three_streams.txt 2>> $O/err.log
rm -rf $O/prof2
cat $O/thin_sweep.txt | tail -40
