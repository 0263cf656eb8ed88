#!/bin/bash
# round 6: loss head with the packed butterfly: tests, per-launch time (rocprofv3), iteration rate
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6hd; export TMPDIR=/tmp
python -m pytest tests/test_closure_gpu.py tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_notebook_gpu.py -x -q -m gpu 2>&1 | tail -4
python -m pytest tests/test_net_gpu.py -x -q -m gpu -k "not end_quality" 2>&1 | tail -3
ROOTD=$(pwd); O=$ROOTD/gpurun_out/r6hd
B="--steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
for cfg in default library; do
( cd /tmp && env DIP_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o trace -- python $ROOTD/bench.py --config $cfg $B > $O/prof_bench.log 2>&1 )
python tools/prof_summary.py $O/prof1 13 2>> $O/err.log | grep -E "loss_head|optimisation steps" | cut -c1-170 | tee -a $O/head.txt
rm -rf $O/prof1
done
rm -f gpurun_out/ab.log
REPS=2 STEPS=150 tools/gpu_ab.sh
