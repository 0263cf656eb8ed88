set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/round.log
: > $L
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "wgrad" --no-header -p no:cacheprovider >> $L 2>&1
timeout 900 python -m pytest tests/test_closure_gpu.py -q -m gpu --no-header -p no:cacheprovider >> $L 2>&1
timeout 900 python -m pytest tests/test_net_gpu.py -q -m gpu -k "golden" --no-header -p no:cacheprovider >> $L 2>&1
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --dump-ops gpurun_out/ops.json --mode eager --no-eager-line 2>&1 | grep '^{"metric"' | cut -c1-200 >> $L
timeout 600 env DIP_LOSS_HEAD_NO_COAL=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --mode eager --no-eager-line 2>&1 | grep '^{"metric"' | cut -c1-200 >> $L
grep -v "^$" $L | tail -30
python - <<'PY'
import json
d=json.load(open('gpurun_out/ops.json'))
for k in sorted(d):
    if 'out' in k or 'skip_conv' in k and k.startswith('wgrad'): print(k, round(d[k]['ms']*1e3,1))
PY
