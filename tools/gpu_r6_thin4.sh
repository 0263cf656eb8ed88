#!/bin/bash
# round 6: conv_thin4 on the matrix pipe: tests, per-launch time by rows per walk (single stream, rocprofv3), iteration rate A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6h; export TMPDIR=/tmp
python -m pytest tests/test_thin_gpu.py -x -q -m gpu -k thin4 2>&1 | tail -15
if [ "${QUICK:-0}" = 1 ]; then exit 0; fi
python -m pytest tests/test_kernels_gpu.py tests/test_small_gpu.py tests/test_group_gpu.py -x -q -m gpu 2>&1 | tail -5
python -m pytest tests/test_net_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "not end_quality" 2>&1 | tail -5
ROOTD=$(pwd); O=$ROOTD/gpurun_out/r6h
B="--steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
for v in "DIP_THIN4_VALU=1" "" "DIP_THIN4_TH=4" "DIP_THIN4_TH=8" "DIP_THIN4_TH=16" "DIP_THIN4_TH=32" "DIP_THIN4_TH=64"; do
  ( cd /tmp && env $v DIP_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o trace -- python $ROOTD/bench.py $B > $O/prof_bench.log 2>&1 )
  echo "== $v" | tee -a $O/thin4_by_th.txt; python tools/prof_summary.py $O/prof1 13 2>> $O/err.log | grep -E "conv_thin4|optimisation steps" | cut -c1-170 | tee -a $O/thin4_by_th.txt
  rm -rf $O/prof1
done
rm -f gpurun_out/ab.log
AB="DIP_THIN4_VALU=1" REPS=${REPS:-3} STEPS=${STEPS:-150} tools/gpu_ab.sh
cp gpurun_out/ab.log $O/ab.log
