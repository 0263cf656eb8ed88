#!/bin/bash
# round 6: more knock-outs (frozen parameters, the first 8 iterations whole): what is still exposed?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6k
B="--steps 150 --warmup 20 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
for rep in 1 2; do
for v in "" "DIP_KNOCKOUT=^conv_fwd:s[0-4]\.skip_conv" "DIP_KNOCKOUT=^upcat:" "DIP_KNOCKOUT=^upb_" "DIP_KNOCKOUT=^(wgrad|wgred):s[234]" "DIP_KNOCKOUT=^(wgrad|wgred):s[0-4]\.skip" "DIP_KNOCKOUT=^bnb_(stats|apply|fin):s0" "DIP_KNOCKOUT=^(conv_fwd|dgrad):s[01]\.up1" "DIP_KNOCKOUT=^(conv_fwd|dgrad):s[01]\.down_a" "DIP_KNOCKOUT=^wgrad:s[01]\.(up|down_b)$"; do
  echo "== $v $(env DIP_BENCH_LR=0 DIP_KNOCKOUT_AFTER=8 $v python bench.py $B 2>/dev/null | grep '^{"metric"' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["final_loss"])')" | tee -a gpurun_out/r6k/ab_knockout2.log
done; done
