#!/bin/bash
# round 6: knock-out experiments -- what would the iteration gain if a family of launches cost nothing? (DIP_KNOCKOUT, results wrong)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6k
rm -f gpurun_out/ab.log
AB="${AB:-DIP_KNOCKOUT=^dgthin: DIP_KNOCKOUT=^wg(rad|red):s[01]\.up1 DIP_KNOCKOUT=^wg(rad|red):s[01]\.down_a DIP_KNOCKOUT=^wgred: DIP_KNOCKOUT=^wgrad:s0\.up$ DIP_KNOCKOUT=^wg(rad|red): DIP_KNOCKOUT=^bn_fin: DIP_KNOCKOUT=^bnb_fin:}" REPS=${REPS:-2} STEPS=${STEPS:-150} tools/gpu_ab.sh
cp gpurun_out/ab.log gpurun_out/r6k/ab_knockout.log
