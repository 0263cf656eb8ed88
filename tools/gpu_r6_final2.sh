#!/bin/bash
# round 6, end: the whole GPU suite on the final code, then the evidence set (tools/gpu_round6_final.sh), one call, one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6f
( time python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r6f/r06_gpu_suite_final.log 2>&1
tail -3 gpurun_out/r6f/r06_gpu_suite_final.log
bash tools/gpu_round6_final.sh
