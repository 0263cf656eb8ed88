#!/bin/bash
# round 6: is the knock-out of the low-resolution bn_fin launches confounded by the data (NaN / zeros -> less power -> faster MFMA kernels)?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6k; export TMPDIR=/tmp
ROOTD=$(pwd); O=$ROOTD/gpurun_out/r6k
B="--steps 150 --warmup 20 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
for v in "" "DIP_KNOCKOUT=^bn_fin:s[234] DIP_KNOCKOUT_AFTER=8" "DIP_KNOCKOUT=^bn_fin:s[234] DIP_KNOCKOUT_AFTER=100000"; do
  echo "== $v"
  env $v python bench.py $B 2>/dev/null | grep '^{"metric"' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["final_loss"], d["per_rank_final_loss_hex"])'
done
B="--steps 30 --warmup 10 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
for v in "" "DIP_KNOCKOUT=^bn_fin:s[234] DIP_KNOCKOUT_AFTER=8"; do
  ( cd /tmp && env $v timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o trace -- python $ROOTD/bench.py $B > $O/prof_bench.log 2>&1 )
  echo "== $v"; python tools/prof_summary.py $O/prof1 40 2>> $O/err.log | cut -c1-170 | head -16
  rm -rf $O/prof1
done
