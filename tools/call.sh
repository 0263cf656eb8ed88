set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/round.log
: > $L
timeout 900 python -m pytest tests/test_net_gpu.py -q -m gpu -k "golden or fresh_seed or default_net_64 or input_gradient" --no-header -p no:cacheprovider -s >> $L 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "upcat or bn_forward or avgpool" --no-header -p no:cacheprovider >> $L 2>&1
grep -v "^$" $L | tail -40
