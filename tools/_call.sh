set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py build > gpurun_out/c20_build.log 2>&1
DIP_TEST_RATIO=2 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider > gpurun_out/c20_ratio2.log 2>&1; echo "ratio2 rc=$?"; grep -E "^FAILED|passed|failed" gpurun_out/c20_ratio2.log | cut -c1-220 | head -40
timeout 600 python -m pytest tests/test_net_gpu.py -q -m gpu -k "fused_batchnorm" --no-header -p no:cacheprovider 2>&1 | tail -2
