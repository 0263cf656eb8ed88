set -u
ROOTD="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOTD"
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4c14
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -2
timeout 900 python -m pytest tests/test_net_gpu.py -q -m gpu -k "not end_quality" --no-header -p no:cacheprovider 2>&1 | tail -2
: > gpurun_out/${T}_ab.log
for rep in 1 2; do
for v in base DIP_WGRAD_NO_BF3=1 DIP_CONV_BF3=0 DIP_CONV_BF3=6; do
  if [ "$v" = base ]; then envs=""; else envs="${v//,/ }"; fi
  line=$(env $envs timeout 300 python bench.py --steps 100 --warmup 20 --mode eager --no-cpu-baseline --no-roofline --no-eager-line 2>gpurun_out/${T}_bench_err.log | grep '^{"metric"' | tail -1)
  echo "$v rep$rep $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["kernel_launches_per_iteration"], d["config"]["final_loss"])' 2>/dev/null)" | tee -a gpurun_out/${T}_ab.log
done
done
