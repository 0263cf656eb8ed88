set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
(rocm-smi --showperflevel --showclocks --showpower --showmaxpower 2>/dev/null | grep -vE "^=|^$" | head -30) > gpurun_out/smi.log
python __graft_entry__.py build > gpurun_out/c1_build.log 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv_wgrad" --no-header -p no:cacheprovider -x > gpurun_out/c1_wgrad_tests.log 2>&1; echo "wgrad tests rc=$?"
tail -3 gpurun_out/c1_wgrad_tests.log
: > gpurun_out/ab.log
AB="DIP_WGRAD_NO_SLIDE=1 DIP_WGRAD_64=1" REPS=3 bash tools/gpu_ab.sh
timeout 300 python tools/wgrad_sweep.py > gpurun_out/c1_wgrad_sweep_slide.log 2>&1
DIP_WGRAD_NO_SLIDE=1 timeout 300 python tools/wgrad_sweep.py > gpurun_out/c1_wgrad_sweep_noslide.log 2>&1
timeout 300 python tools/find_copies.py fused > gpurun_out/c1_copies.log 2>&1
head -40 gpurun_out/c1_copies.log
