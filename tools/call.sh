set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export SKIP_TESTS=0 DO_PROF=1 DO_PROF2=1 DO_PMC=1 EXTRA_BENCH="sr kate library snail" BENCH_INSTANCES=8
bash tools/gpu_round.sh
