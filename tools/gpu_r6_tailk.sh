#!/bin/bash
# round 6: conv_bf3 with the packed K of a 4-channel last chunk: tests, per-launch times (rocprofv3, single stream), iteration rate
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6tk; export TMPDIR=/tmp
python -m pytest tests/test_bf3_gpu.py -x -q -m gpu 2>&1 | tail -4
if [ "${QUICK:-0}" = 1 ]; then exit 0; fi
python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_net_gpu.py tests/test_fullsize_gpu.py tests/test_closure_gpu.py -x -q -m gpu -k "not end_quality" 2>&1 | tail -4
ROOTD=$(pwd); O=$ROOTD/gpurun_out/r6tk
B="--steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
( cd /tmp && env DIP_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o trace -- python $ROOTD/bench.py $B > $O/prof_bench.log 2>&1 )
python tools/prof_summary.py $O/prof1 13 2>> $O/err.log | cut -c1-170 | head -12 | tee $O/stats.txt
rm -rf $O/prof1
python bench.py --steps 100 --warmup 10 --mode eager --no-cpu-baseline --no-eager-line --dump-ops $O/ops.json 2>/dev/null | grep '^{"metric"' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["largest_layer"])'
rm -f gpurun_out/ab.log
REPS=2 STEPS=150 tools/gpu_ab.sh
