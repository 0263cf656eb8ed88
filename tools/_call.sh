set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py build > gpurun_out/c23_build.log 2>&1
timeout 900 python -m pytest tests/test_net_gpu.py -q -m gpu -k "golden" --no-header -p no:cacheprovider > gpurun_out/c23_golden.log 2>&1; echo "golden rc=$?"; tail -5 gpurun_out/c23_golden.log | cut -c1-300
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "upcat or bn_" --no-header -p no:cacheprovider 2>&1 | tail -2
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
