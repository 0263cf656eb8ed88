set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export SKIP_BENCH=1
bash tools/gpu_round.sh
