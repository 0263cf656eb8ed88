#!/bin/bash
# round 6, second session: the streaming weight-gradient tail (wgrad_tail.hip): tests, per-launch times by NCB, A/B in the iteration
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6t; export TMPDIR=/tmp
python -m pytest tests/test_bf3_gpu.py -x -q -m gpu -k "wgrad_bf3" 2>&1 | tail -5
rm -f gpurun_out/ab.log
AB="${AB:-DIP_WGRAD_TAIL_OLD=1 DIP_WGRAD_TAIL_NCB=1 DIP_WGRAD_TAIL_NCB=4}" REPS=${REPS:-3} STEPS=${STEPS:-150} tools/gpu_ab.sh
cp gpurun_out/ab.log gpurun_out/r6t/ab_tail.log
ROOTD=$(pwd); O=$ROOTD/gpurun_out/r6t
B="--steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
for v in "" "DIP_WGRAD_TAIL_NCB=1" "DIP_WGRAD_TAIL_NCB=4"; do
  ( cd /tmp && env $v DIP_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o trace -- python $ROOTD/bench.py $B > $O/prof_bench.log 2>&1 )
  echo "== $v"; python tools/prof_summary.py $O/prof1 13 2>> $O/err.log | grep -E "wgrad_tail|conv_wgrad_kernel<3, 1, 9, 1, true|optimisation steps" | cut -c1-170
  rm -rf $O/prof1
done
