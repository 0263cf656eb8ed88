#!/bin/bash
# A/B of environment switches on ONE box: interleaved bench.py runs (eager, no roofline/cpu legs).
#   AB="DIP_X=1 DIP_Y=1" REPS=3 STEPS=100 tools/gpu_ab.sh     ("base" = no switch is always included)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/ab.log
for rep in $(seq 1 ${REPS:-3}); do
  for v in base ${AB:-}; do
    if [ "$v" = base ]; then envs=""; else envs="$v"; fi
    line=$(env ${envs//,/ } timeout 300 python bench.py --steps ${STEPS:-100} --warmup 20 --mode ${MODE:-eager} --no-cpu-baseline --no-roofline --no-eager-line ${BENCH_ARGS:-} 2>/dev/null | grep '^{"metric"' | tail -1)
    echo "$v rep$rep $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>/dev/null)" | tee -a $OUT
  done
done
