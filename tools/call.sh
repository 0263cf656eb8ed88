set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/round.log
: > $L
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv_forward or conv_dgrad" --no-header -p no:cacheprovider >> $L 2>&1
for e in "DIP_CONV_SWZ_OLD=1" "DIP_X=1" "DIP_CONV_SWZ_OLD=1" "DIP_X=1"; do
  echo "== bench $e" >> $L
  timeout 600 env $e python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-eager-line --mode eager 2>&1 | grep '^{"metric"' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], 'it/s; dominant', r['achieved'], 'TF frac', r['frac'], 's0.up fwd', r['largest_layer']['us'], 'us; all conv', r['all_conv_launches'])" >> $L
done
grep -v "^$" $L | tail -20
