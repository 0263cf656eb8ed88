set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python tools/drift_probe.py hip 600 128 > gpurun_out/drift_hip.json 2> gpurun_out/drift_hip.err; echo rc=$?
DIP_CONV_PLAN_WGS=256 DIP_WGRAD_NO_SMALL_PLAN=1 python tools/drift_probe.py hip 600 128 > gpurun_out/drift_hip_arm3.json 2>> gpurun_out/drift_hip.err; echo rc=$?
