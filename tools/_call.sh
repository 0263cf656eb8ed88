set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py build > gpurun_out/c8_build.log 2>&1
run() { local name=$1; shift; local t0=$SECONDS; timeout 1500 "$@" > gpurun_out/c8_$name.log 2>&1; echo "$name rc=$? $((SECONDS-t0))s"; tail -3 gpurun_out/c8_$name.log | cut -c1-300; }
run smoke python __graft_entry__.py smoke
run closure python -m pytest tests/test_closure_gpu.py -q -m gpu --no-header -p no:cacheprovider
: > gpurun_out/ab.log
AB="DIP_DEFER_WGRAD=-1 DIP_DEFER_WGRAD=1 DIP_DEFER_WGRAD=0 DIP_SIDE_MIN_PIXELS=0 DIP_SIDE_MIN_PIXELS=65536 DIP_TWO_STREAMS=0" REPS=2 bash tools/gpu_ab.sh
MODE=graph AB="DIP_DEFER_WGRAD=-1" REPS=1 bash tools/gpu_ab.sh
