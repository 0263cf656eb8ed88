set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export SKIP_BENCH=1 TESTS="tests/test_kernels_gpu.py tests/test_net_gpu.py::test_golden_reference_vectors tests/test_net_gpu.py::test_against_oracle_fresh_seed"
bash tools/gpu_round.sh
timeout 300 python __graft_entry__.py smoke >> gpurun_out/round.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/round.log
