#!/bin/bash
# Round 5, second GPU call: the ping-pong weight gradient (bit-identity with the 4-wave form, timing, A/B), the 64-column tile as default
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5b; mkdir -p $O; export TMPDIR=/tmp
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.log; }
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; stamp "build rc=$?"
timeout 400 python -m pytest tests/test_bf3_gpu.py -m gpu -q -x > $O/bf3_tests.log 2>&1; stamp "bf3 tests rc=$? $(tail -n 1 $O/bf3_tests.log)"
timeout 120 python tools/wgrad_bf3_time.py > $O/wgrad_time_pingpong.log 2>&1; stamp "wgrad time rc=$?"
DIP_WGRAD_BF3_V1=1 timeout 120 python tools/wgrad_bf3_time.py > $O/wgrad_time_v1.log 2>&1; stamp "wgrad time v1 rc=$?"
AB="DIP_WGRAD_BF3_V1=1" REPS=2 STEPS=100 timeout 300 tools/gpu_ab.sh > $O/ab.log 2>&1; stamp "ab done"
B="--steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
( cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof1 -o trace -- env DIP_TWO_STREAMS=0 python $OLDPWD/bench.py $B > $OLDPWD/$O/prof1.log 2>&1 )
python tools/prof_summary.py $O/prof1 13 > $O/kernel_stats_single_stream.txt 2>> $O/prof1.log
rm -rf $O/prof1; stamp "prof done"
timeout 200 python bench.py --steps 100 --warmup 10 > $O/bench.log 2>&1; grep '^{"metric"' $O/bench.log | tail -1 > $O/bench_line.json; stamp "bench done"
