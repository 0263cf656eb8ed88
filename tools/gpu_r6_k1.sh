#!/bin/bash
# round 6: 1x1 layers on the bf16 pipe (conv_bf3_k1_kernel): tests, per-launch times (single stream, rocprofv3), iteration rate A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6k1; export TMPDIR=/tmp
python -m pytest tests/test_bf3_gpu.py -x -q -m gpu -k "1x1" 2>&1 | tail -15
if [ "${QUICK:-0}" = 1 ]; then exit 0; fi
python -m pytest tests/test_kernels_gpu.py tests/test_bf3_gpu.py tests/test_group_gpu.py -x -q -m gpu 2>&1 | tail -5
python -m pytest tests/test_net_gpu.py tests/test_fullsize_gpu.py tests/test_closure_gpu.py -x -q -m gpu -k "not end_quality" 2>&1 | tail -5
ROOTD=$(pwd); O=$ROOTD/gpurun_out/r6k1
B="--steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
for v in "DIP_CONV_BF3_NO_1X1=1" ""; do
  ( cd /tmp && env $v DIP_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o trace -- python $ROOTD/bench.py $B > $O/prof_bench.log 2>&1 )
  echo "== $v" | tee -a $O/k1_kernels.txt; python tools/prof_summary.py $O/prof1 13 2>> $O/err.log | grep -E "conv_bf3_k1|conv1x1_res|optimisation steps" | cut -c1-170 | tee -a $O/k1_kernels.txt
  rm -rf $O/prof1
done
rm -f gpurun_out/ab.log
AB="DIP_CONV_BF3_NO_1X1=1" REPS=${REPS:-3} STEPS=${STEPS:-150} tools/gpu_ab.sh
cp gpurun_out/ab.log $O/ab.log
