set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOTD=$(pwd); O=$ROOTD/gpurun_out; T=r04
P="--steps 3 --warmup 2 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
PRE="env LD_PRELOAD=$ROOTD/deep-image-prior_amd/lib/libdip_hip.so DIP_TWO_STREAMS=0"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_cal/a -o pmc -- $ROOTD/tools/ubench/bin/mfma_peak > /dev/null 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_cal/b -o pmc -- $ROOTD/tools/ubench/bin/bf16x9 > /dev/null 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_MFMA -o pmc -- $PRE python $ROOTD/bench.py $P > /dev/null 2>&1 )
python tools/pmc_mfma.py $O/pmc_cal $O/pmc_MFMA $O/${T}_pmc_mfma.json > $O/${T}_rocprofv3_pmc_MFMA.txt
rm -rf $O/pmc_cal $O/pmc_MFMA
head -24 $O/${T}_rocprofv3_pmc_MFMA.txt
