set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/round.log
: > $L
for e in "DIP_CONV_PLAN_WGS=512" "DIP_CONV_PLAN_WGS=384" "DIP_CONV_PLAN_WGS=256" "DIP_CONV_PLAN_WGS=512" "DIP_CONV_PLAN_WGS=384" "DIP_CONV_PLAN_WGS=448"; do
  echo "== bench $e" >> $L
  timeout 600 env $e python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-eager-line --mode eager 2>&1 | grep '^{"metric"' | cut -c1-140 >> $L
done
for e in "DIP_CONV_PLAN_WGS=384" "DIP_CONV_PLAN_WGS=256"; do
  echo "== snail $e" >> $L
  timeout 600 env $e python bench.py --config snail --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-eager-line 2>&1 | grep '^{"metric"' | cut -c1-140 >> $L
done
grep -v "^$" $L | tail -60
