set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py build > gpurun_out/c21_build.log 2>&1
timeout 2400 python -m pytest tests/test_net_gpu.py -q -m gpu -k "end_quality" --no-header -p no:cacheprovider -s > gpurun_out/c21_eq.log 2>&1; echo "eq rc=$?"
grep -E "psnr_gt|loss:|passed|failed" gpurun_out/c21_eq.log | cut -c1-300
