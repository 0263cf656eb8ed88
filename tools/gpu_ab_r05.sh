#!/bin/bash
# interleaved A/B of THIS tree against the round-5 tree (git worktree _r05/, built in place) on ONE box
#   CFGS="default library" REPS=3 tools/gpu_ab_r05.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; O=gpurun_out/ab_r05.log; : > $O
for rep in $(seq 1 ${REPS:-3}); do
 for tree in . _r05; do
  for cfg in ${CFGS:-default}; do
    line=$(cd $tree && timeout 300 python bench.py --config $cfg --steps ${STEPS:-100} --warmup 20 --mode eager --no-cpu-baseline --no-roofline --no-eager-line ${BENCH_ARGS:-} 2>/dev/null | grep '^{"metric"' | tail -1)
    echo "$cfg tree=$tree rep$rep $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["kernel_launches_per_iteration"], d.get("build_id"))' 2>/dev/null)" | tee -a $O
  done
 done
done
