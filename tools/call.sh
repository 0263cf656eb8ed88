set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/round.log
: > $L
timeout 120 python __graft_entry__.py build >> $L 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv_forward or conv_dgrad" --no-header -p no:cacheprovider >> $L 2>&1
timeout 900 python -m pytest tests/test_net_gpu.py -q -m gpu -k "golden or fresh_seed or default_net_64" --no-header -p no:cacheprovider >> $L 2>&1
for e in "DIP_CONV_NO_S2DMA=1" "DIP_X=1" "DIP_CONV_NO_S2DMA=1" "DIP_X=1"; do
  echo "== bench $e" >> $L
  timeout 600 env $e python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-eager-line --mode eager 2>&1 | grep '^{"metric"' | cut -c1-140 >> $L
done
timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --dump-ops gpurun_out/ops.json --mode eager --no-eager-line 2>&1 | grep '^{"metric"' | cut -c1-1500 >> $L
grep -v "^$" $L | tail -60
