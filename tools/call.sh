set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/round.log
: > $L
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "upcat or bn_forward or avgpool" --no-header -p no:cacheprovider >> $L 2>&1
timeout 900 python -m pytest tests/test_net_gpu.py -q -m gpu -k "golden or fresh_seed or default_net_64" --no-header -p no:cacheprovider >> $L 2>&1
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -k "upcat" --no-header -p no:cacheprovider >> $L 2>&1
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --dump-ops gpurun_out/ops.json --mode eager --no-eager-line 2>&1 | grep '^{"metric"' | cut -c1-200 >> $L
grep -v "^$" $L | tail -30
python - <<'PY'
import json
d=json.load(open('gpurun_out/ops.json'))
for k in sorted(d):
    if k.startswith('upcat') or k.startswith('upb_stats'): print(k, round(d[k]['ms']*1e3,1))
PY
