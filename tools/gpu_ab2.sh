#!/bin/bash
# interleaved A/B of environment switches over several configurations on ONE box:
#   AB="DIP_X=1 DIP_Y=1,DIP_Z=2" CFGS="default library" REPS=3 tools/gpu_ab2.sh      ("base" is always included)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; O=gpurun_out/${OUT:-ab2.log}; : > $O
for rep in $(seq 1 ${REPS:-3}); do
 for v in base ${AB:-}; do
  if [ "$v" = base ]; then envs=""; else envs="${v//,/ }"; fi
  for cfg in ${CFGS:-default}; do
    line=$(env $envs timeout 300 python bench.py --config $cfg --steps ${STEPS:-100} --warmup 20 --mode ${MODE:-eager} --no-cpu-baseline --no-roofline --no-eager-line ${BENCH_ARGS:-} 2>/dev/null | grep '^{"metric"' | tail -1)
    echo "$cfg $v rep$rep $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["kernel_launches_per_iteration"])' 2>/dev/null)" | tee -a $O
  done
 done
done
