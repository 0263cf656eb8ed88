#!/bin/bash
# round 6, second session: does capping wgrad_bf3's grid (free CUs for the main stream's chain) pay?  A/B + one timeline
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6w; export TMPDIR=/tmp
rm -f gpurun_out/ab.log
AB="${AB:-DIP_WGRAD_BF3_WGS=240 DIP_WGRAD_BF3_WGS=224 DIP_WGRAD_BF3_WGS=192}" REPS=${REPS:-3} STEPS=${STEPS:-150} tools/gpu_ab.sh
cp gpurun_out/ab.log gpurun_out/r6w/ab_wgs.log
if [ -n "${TL:-}" ]; then
  ROOTD=$(pwd); O=$ROOTD/gpurun_out/r6w
  B="--steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
  ( cd /tmp && env $TL timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof2 -o trace -- python $ROOTD/bench.py $B > $O/prof_bench.log 2>&1 )
  python tools/prof_timeline.py $O/prof2 3 > $O/timeline_three_streams_wgs.txt 2>> $O/err.log
  rm -rf $O/prof2
  head -8 $O/timeline_three_streams_wgs.txt | cut -c1-160
fi
