#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6x; export TMPDIR=/tmp
python -m pytest tests/test_thin_gpu.py -x -q -m gpu -k thin4 2>&1 | tail -2
DIP_DEFER_WGRAD=-1 python tools/race_loop.py 4000 2>&1 | tail -1
python tools/race_loop.py 2000 2>&1 | tail -1
ROOTD=$(pwd); O=$ROOTD/gpurun_out/r6x
B="--steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
( cd /tmp && env DIP_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o trace -- python $ROOTD/bench.py $B > $O/prof_bench.log 2>&1 )
python tools/prof_summary.py $O/prof1 13 2>> $O/err.log | grep -E "conv_thin4|optimisation steps" | cut -c1-170
rm -rf $O/prof1
rm -f gpurun_out/ab.log; REPS=2 STEPS=150 tools/gpu_ab.sh
