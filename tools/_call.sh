set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py build > gpurun_out/c13_build.log 2>&1
timeout 300 python tools/wgrad_sweep.py 2>&1 | grep -E "^13[0-9]>|^128>128 k3s1 (512|256)" | head -8
echo ---- no phase 2
DIP_WGRAD_DEBUG_NO_PHASE2=1 timeout 300 python tools/wgrad_sweep.py 2>&1 | grep -E "^13[0-9]>|^128>128 k3s1 (512|256)" | head -8
