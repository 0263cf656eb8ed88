set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
B="python bench.py --steps 100 --warmup 10 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
run() { echo "== $*"; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['final_loss'])"; }
run X=1
run DIP_BF3_LDSB=1
run X=1
run DIP_BF3_LDSB=1
timeout 900 python -m pytest tests/test_bf3_gpu.py tests/test_small_gpu.py -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_net_gpu.py -x -q -k "iter1 or tiny or golden" 2>&1 | tail -3
