#!/bin/bash
# round 6: knock-outs with the data kept real: learning rate 0 (parameters frozen), the first 8 iterations whole, then the launches left out
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6k
B="--steps 150 --warmup 20 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
for rep in 1 2; do
for v in "" "DIP_KNOCKOUT=^bn_fin:" "DIP_KNOCKOUT=^bn_fin:s[234]" "DIP_KNOCKOUT=^bn_fin:s[01]" "DIP_KNOCKOUT=^bnb_fin:" "DIP_KNOCKOUT=^(bnb_fin|bn_fin):" "DIP_KNOCKOUT=^wgred:" "DIP_KNOCKOUT=^(bnb_|bn_fin)" "DIP_KNOCKOUT=^(conv_fwd|dgrad.?|upcat|bnb_one):s[234]"; do
  echo "== $v $(env DIP_BENCH_LR=0 DIP_KNOCKOUT_AFTER=8 $v python bench.py $B 2>/dev/null | grep '^{"metric"' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["final_loss"])')" | tee -a gpurun_out/r6k/ab_knockout_lr0.log
done; done
