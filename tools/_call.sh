set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py build > gpurun_out/c15_build.log 2>&1
run() { local name=$1; shift; local t0=$SECONDS; timeout 1500 "$@" > gpurun_out/c15_$name.log 2>&1; echo "$name rc=$? $((SECONDS-t0))s"; tail -4 gpurun_out/c15_$name.log | cut -c1-400; }
run kern python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv_forward or conv_dgrad or fused" --no-header -p no:cacheprovider -x
run smoke python __graft_entry__.py smoke
run full python -m pytest tests/test_fullsize_gpu.py -q -m gpu --no-header -p no:cacheprovider -x -k "default_net or kernel or layer"
: > gpurun_out/ab.log
AB="DIP_CONV_NO_RES1X1=1" REPS=3 bash tools/gpu_ab.sh
python bench.py --steps 50 --warmup 10 --mode eager --no-cpu-baseline --no-eager-line --dump-ops gpurun_out/ops.json > gpurun_out/c15_bench.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/ops.json'))
for k in ("conv_fwd:s0.up1","dgrad:s0.up1","conv_fwd:s1.up1","dgrad:s1.up1","conv_fwd:s0.up","dgrad:s0.up","wgrad:s0.up"):
    print(k, d.get(k))
PY
