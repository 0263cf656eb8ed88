#!/bin/bash
# round 6: what bounds the library config?  eager / hipGraph x stream forms, and knock-outs (GPU work removed: host-bound if the rate stays)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6l
B="--config library --steps 150 --warmup 20 --no-cpu-baseline --no-roofline --no-eager-line"
run() { echo "== $* $(env "$@" python bench.py $B --mode $MODE 2>/dev/null | grep '^{"metric"' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["final_loss"])')" | tee -a gpurun_out/r6l/library_forms.txt; }
for rep in 1 2; do
MODE=eager run X=0
MODE=graph run X=0
MODE=eager run DIP_TWO_STREAMS=0
MODE=graph run DIP_TWO_STREAMS=0
MODE=eager run DIP_KNOCKOUT=^wg DIP_KNOCKOUT_AFTER=8 DIP_BENCH_LR=0
MODE=eager run DIP_KNOCKOUT=^bn DIP_KNOCKOUT_AFTER=8 DIP_BENCH_LR=0
MODE=eager run "DIP_KNOCKOUT=^(wg|bn|dg|conv)" DIP_KNOCKOUT_AFTER=8 DIP_BENCH_LR=0
done
