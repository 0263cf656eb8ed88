set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/round.log
: > $L
timeout 120 python __graft_entry__.py build >> $L 2>&1
for e in "DIP_CONV_NO_PHASE=1" "DIP_X=1" "DIP_CONV_PHASE_KSPLIT=1" "DIP_CONV_PHASE_KSPLIT=2"; do
  timeout 200 env $e python tools/dgrad2_sweep.py 2>&1 | grep -v amdgpu.ids >> $L
done
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "dgrad" --no-header -p no:cacheprovider >> $L 2>&1
timeout 600 python -m pytest tests/test_net_gpu.py -q -m gpu -k "golden or fresh_seed" --no-header -p no:cacheprovider >> $L 2>&1
for e in "DIP_CONV_NO_PHASE=1" "DIP_X=1" "DIP_CONV_NO_PHASE=1" "DIP_X=1"; do
  echo "== bench $e" >> $L
  timeout 600 env $e python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-eager-line --mode eager 2>&1 | grep '^{"metric"' | cut -c1-140 >> $L
done
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --dump-ops gpurun_out/ops.json >> $L 2>&1
grep -v "^$" $L | tail -60
