set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/round.log
: > $L
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "wgrad" --no-header -p no:cacheprovider >> $L 2>&1
timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -k "appa_conv" --no-header -p no:cacheprovider >> $L 2>&1
for e in "DIP_WGRAD_NO_DMA_DY=1" "DIP_X=1"; do
  echo "== sweep $e" >> $L
  timeout 300 env $e python tools/wgrad_sweep.py 2>&1 | grep -E "k3s1 (512|256|128):" >> $L
done
for e in "DIP_WGRAD_NO_DMA_DY=1" "DIP_X=1" "DIP_WGRAD_NO_DMA_DY=1" "DIP_X=1"; do
  echo "== bench $e" >> $L
  timeout 600 env $e python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-eager-line --mode eager 2>&1 | grep '^{"metric"' | cut -c1-140 >> $L
done
grep -v "^$" $L | tail -40
