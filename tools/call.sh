set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/round.log
: > $L
for e in "DIP_WGRAD_NO_64=1" "DIP_WGRAD64_WGS=128" "DIP_WGRAD64_WGS=192" "DIP_WGRAD64_WGS=64" "DIP_WGRAD_NO_64=1" "DIP_WGRAD64_WGS=128" "DIP_WGRAD64_WGS=256"; do
  echo "== bench $e" >> $L
  timeout 600 env $e python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-eager-line --mode eager 2>&1 | grep '^{"metric"' | cut -c1-140 >> $L
done
grep -v "^$" $L | tail -120
