set -u
ROOTD="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOTD"
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py build > gpurun_out/c30_build.log 2>&1
# SQ counters in a pass of their own (no trace domain besides --kernel-trace); library preloaded (counter service vs dlopen)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $ROOTD/gpurun_out/pmc_SQ -o pmc -- env DIP_TWO_STREAMS=0 LD_PRELOAD=$ROOTD/deep-image-prior_amd/lib/libdip_hip.so python $ROOTD/bench.py --steps 3 --warmup 2 --mode eager --no-cpu-baseline --no-roofline --no-eager-line > $ROOTD/gpurun_out/c30_sq.log 2>&1 ); echo "sq rc=$?"
python tools/pmc_sq.py gpurun_out/pmc_SQ > gpurun_out/pmc_SQ_summary.txt 2>> gpurun_out/c30_sq.log
rm -rf gpurun_out/pmc_SQ
head -12 gpurun_out/pmc_SQ_summary.txt | cut -c1-160
