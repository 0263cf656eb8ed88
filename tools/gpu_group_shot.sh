#!/bin/bash
# Grouped multi-instance execution (dip_group.GroupedFits) on a GPU box in ONE short gpurun call, most important first:
# parity tests of the grouped path, bench lines of the small configs x 8 instances (grouped vs one graph per fit), the
# headline bench as a regression check of the solo path, smoke(), a kernel trace of the grouped snail iteration.
#   gpurun --timeout 560 -- 'bash tools/gpu_group_shot.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOTD=$(pwd); O=$ROOTD/gpurun_out; T=r04g
LOG=$O/${T}_shot.log; : > $LOG
CUR=$O/.cur.log
run() { echo "=== $* ===" | tee -a $LOG; local t0=$SECONDS; timeout "${TMO:-120}" "$@" > $CUR 2>&1; local rc=$?; cat $CUR >> $LOG; echo "--- rc=$rc ($((SECONDS-t0)) s, t=$SECONDS) ---" | tee -a $LOG; }
line() { grep '^{"metric"' $CUR | tail -1 > $O/$1; python - $O/$1 <<'EOF' | tee -a $LOG
import json, sys
try:
    o = json.loads(open(sys.argv[1]).read())
    print("LINE", sys.argv[1].split("/")[-1], o["value"], "it/s |", o["config"].get("reported_mode"), "| loss", o["config"].get("final_loss"),
          "| other:", json.dumps(o.get("other_mode")))
except Exception as e:
    print("LINE", sys.argv[1], "unreadable:", e)
EOF
}
X8="--instances 8 --steps 60 --warmup 10 --mode graph --no-cpu-baseline --no-roofline --no-eager-line"
TMO=240 run python __graft_entry__.py build
TMO=240 run python -m pytest tests/test_group_gpu.py -q -m gpu -n 1 --timeout 90 --no-header -p no:cacheprovider -rA
grep -E "^(PASSED|FAILED|ERROR)|passed|failed" $LOG | tail -40 > $O/${T}_group_tests.txt
TMO=100 run env DIP_TWO_STREAMS=0 python bench.py --config snail $X8
line ${T}_bench_snail_x8_single_stream.json
TMO=100 run python bench.py --config snail $X8
line ${T}_bench_snail_x8.json
TMO=120 run python bench.py --steps 60 --warmup 10 --mode eager --no-cpu-baseline --no-eager-line
line ${T}_bench_line_after_group_refactor.json
TMO=150 run env DIP_TWO_STREAMS=0 python bench.py --config library $X8
line ${T}_bench_library_x8_single_stream.json
TMO=120 run python __graft_entry__.py smoke
( cd /tmp && TMO=120 run rocprofv3 --kernel-trace --stats -d $O/profg -o trace -- env DIP_TWO_STREAMS=0 python $ROOTD/bench.py --config snail --instances 8 --group native --mode eager --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-eager-line )
python tools/prof_summary.py $O/profg 8 > $O/${T}_rocprofv3_kernel_stats_snail_x8_grouped.txt 2>> $LOG
rm -rf $O/profg
TMO=150 run python bench.py --config library $X8
line ${T}_bench_library_x8.json
TMO=200 run python -m pytest tests/test_closure_gpu.py tests/test_small_gpu.py -q -m gpu -n 2 --no-header -p no:cacheprovider
TMO=150 run python bench.py --instances 2 --steps 40 --warmup 8 --mode auto --no-cpu-baseline --no-roofline --no-eager-line
line ${T}_bench_default_x2.json
echo "=========== summary"; grep -E "^LINE|passed|failed|^FAILED|^ERROR|--- rc=" $LOG | tail -45
