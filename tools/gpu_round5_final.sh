#!/bin/bash
# Evidence set of the round's FINAL code in one short gpurun call: counter passes (HBM traffic, MFMA utilisation + its
# calibration), the bench line that carries them, rocprofv3 kernel stats of the same command (single stream / three streams).
#   gpurun --timeout 400 -- 'bash tools/gpu_round5_final.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOTD=$(pwd); O=$ROOTD/gpurun_out; T=r05
LOG=$O/${T}_final.log; : > $LOG
run() { echo "=== $* ===" | tee -a $LOG; local t0=$SECONDS; timeout "${TMO:-120}" "$@" >> $LOG 2>&1; echo "--- rc=$? ($((SECONDS-t0)) s, t=$SECONDS) ---" | tee -a $LOG; }
P="--steps 3 --warmup 2 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
PRE="env LD_PRELOAD=$ROOTD/deep-image-prior_amd/lib/libdip_hip.so DIP_TWO_STREAMS=0"
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && TMO=100 run rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_$ctr -o pmc -- $PRE python $ROOTD/bench.py $P )
done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/${T}_pmc_traffic.json 2>> $LOG
python tools/pmc_summary.py $O/pmc_FETCH_SIZE > $O/${T}_rocprofv3_pmc_FETCH_SIZE.txt 2>> $LOG
python tools/pmc_summary.py $O/pmc_WRITE_SIZE > $O/${T}_rocprofv3_pmc_WRITE_SIZE.txt 2>> $LOG
( cd /tmp && TMO=60 run rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_cal/a -o pmc -- $ROOTD/tools/ubench/bin/mfma_peak )
( cd /tmp && TMO=60 run rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_cal/b -o pmc -- $ROOTD/tools/ubench/bin/bf16x9 )
( cd /tmp && TMO=100 run rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_MFMA -o pmc -- $PRE python $ROOTD/bench.py $P )
python tools/pmc_mfma.py $O/pmc_cal $O/pmc_MFMA $O/${T}_pmc_mfma.json > $O/${T}_rocprofv3_pmc_MFMA.txt 2>> $LOG
cp $O/${T}_pmc_traffic.json $O/${T}_pmc_mfma.json profiles/ 2>> $LOG       # (bench.py reads them: same call, same box)
TMO=200 run python bench.py --steps 100 --warmup 10 --dump-ops $O/${T}_ops.json
grep '^{"metric"' $LOG | tail -1 > $O/${T}_bench_line.json
B="--steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
( cd /tmp && TMO=100 run rocprofv3 --kernel-trace --stats -d $O/prof1 -o trace -- env DIP_TWO_STREAMS=0 python $ROOTD/bench.py $B )
python tools/prof_summary.py $O/prof1 13 > $O/${T}_rocprofv3_kernel_stats_single_stream.txt 2>> $LOG
python tools/prof_timeline.py $O/prof1 3 > $O/${T}_timeline_single_stream.txt 2>> $LOG
( cd /tmp && TMO=100 run rocprofv3 --kernel-trace --stats -d $O/prof2 -o trace -- python $ROOTD/bench.py $B )
python tools/prof_summary.py $O/prof2 13 > $O/${T}_rocprofv3_kernel_stats_three_streams.txt 2>> $LOG
python tools/prof_timeline.py $O/prof2 3 > $O/${T}_timeline_three_streams.txt 2>> $LOG
rm -rf $O/prof1 $O/prof2 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_cal $O/pmc_MFMA
python - <<'PY'
import json
o = json.load(open("gpurun_out/r05_bench_line.json"))
r = o["roofline"]
print("LINE", o["value"], "it/s", o["ms_per_step"], "ms | frac", r["frac"], "achieved", r["achieved"], "peak", r["peak"], "| pmc util", r.get("mfma_util_pmc"), "clock", r.get("clock_ghz_pmc"),
      "frac_from_pmc", r.get("frac_from_pmc"), "| traffic", r.get("traffic"), "| wgrad", o["roofline_wgrad"]["frac"], "| 3x3 all", o["roofline_conv3x3_all"]["frac"],
      "| hbm", o["roofline_hbm"]["frac"], "| cpu", o["cpu_baseline"]["value"] if o.get("cpu_baseline") else None, "| power", (o.get("timed_region_power") or {}).get("power_w_mean"))
PY
grep -E "rc=" $LOG | tail -12
