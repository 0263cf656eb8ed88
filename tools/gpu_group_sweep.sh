#!/bin/bash
# How far does grouping go?  Instances per launch list, per configuration.   gpurun --timeout 300 -- 'bash tools/gpu_group_sweep.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=$(pwd)/gpurun_out; T=r04g; LOG=$O/${T}_sweep.log; : > $LOG; CUR=$O/.cur.log
B="--group native --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-eager-line"
one() { local tag=$1; shift; echo "=== $* ===" >> $LOG; local t0=$SECONDS; timeout 100 "$@" > $CUR 2>&1; local rc=$?; cat $CUR >> $LOG
  grep '^{"metric"' $CUR | tail -1 | python -c "
import json,sys
try:
    o=json.loads(sys.stdin.read()); print('$tag', o['value'], 'it/s', o['ms_per_step'], 'ms/step |', o['config'].get('reported_mode'), '| other', json.dumps(o.get('other_mode')))
except Exception as e: print('$tag', 'FAILED rc=$rc', e)
" | tee -a $O/${T}_sweep.txt; echo "--- rc=$rc ($((SECONDS-t0)) s)" >> $LOG; }
: > $O/${T}_sweep.txt
for n in 16 32; do one snail_x$n env DIP_TWO_STREAMS=0 python bench.py --config snail --instances $n --mode graph $B; done
one snail_x8_planwgs64 env DIP_TWO_STREAMS=0 DIP_CONV_PLAN_WGS=64 python bench.py --config snail --instances 8 --mode graph $B
one library_x16 env DIP_TWO_STREAMS=0 python bench.py --config library --instances 16 --mode graph $B
one library_x4 env DIP_TWO_STREAMS=0 python bench.py --config library --instances 4 --mode graph $B
for n in 3 4; do one default_x$n python bench.py --instances $n --mode eager $B; done
one default_x4_graph python bench.py --instances 4 --mode graph $B
one kate_x4 python bench.py --config kate --instances 4 --mode eager $B
cat $O/${T}_sweep.txt
